#!/usr/bin/env python3
"""Development aid: per-slot cycle breakdown of the team-pipelined blend kernel (needs a library built with
-DEGX_LBS_TIMING: `make -C egogen_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DEGX_LBS_TIMING"`)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import synth, _lib
from egogen_amd.body_model import BodyModelHandle, SdfScene
lib = _lib.load()
A, T = 512, 20
bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
g = torch.Generator().manual_seed(0)
xb = (torch.randn(A * T, 93, generator=g) * 0.2).cuda(); xb[:, 2] += 1
betas = torch.randn(A, 10, generator=g).cuda()
R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
raw = C.CDLL(_lib.LIB_PATH)
for mode in (1, 2):
    lib.egx_lbs_set_blend_mode(mode)
    for name, kw in (("picks", {}), ("picks+sdf", dict(sdf=scene, R0=R0, T0=T0))):
        out = {}
        for _ in range(3):
            h.forward(xb, betas, T, out=out, **kw)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        raw.egx_lbs_timing_read(buf, 1)
        h.forward(xb, betas, T, out=out, **kw)
        torch.cuda.synchronize()
        raw.egx_lbs_timing_read(buf, 0)
        nf, nm = max(buf[6], 1), max(buf[3], 1)
        nslots = nf + nm  # wave-slots with work (fetch or mfma)
        print(f"mode {mode} {name}: mfma-slot vmcnt wait {buf[0]/nm:.0f}, fetch-slot work {buf[1]/nf:.0f} | mfma phase (incl. wait) {buf[2]/nm:.0f} | "
              f"barrier wait: mfma team {buf[4]/nm:.0f}, fetch team {buf[5]/nm:.0f} | slot total {buf[7]/(2*nm):.0f} | "
              f"epilogue per wave-item {buf[8]/(nm/ (15 if mode==1 else 10)):.0f}  (wave-slots: {nm} mfma, {nf} fetch)")
        ne = max(buf[12], 1)
        print(f"      epilogue parts per wave-item: skinning {buf[9]/ne:.0f}, sdf brackets+queue {buf[10]/ne:.0f}, flush+counters {buf[11]/ne:.0f}")
