#!/usr/bin/env python3
"""Development aid: cycle breakdown of the fused LBS kernel per wave and work item (library built with -DEGX_LBS_TIMING)."""
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
for mode in (3, 2):
    lib.egx_lbs_set_blend_mode(mode)
    kw = dict(sdf=scene, R0=R0, T0=T0)
    out = {}
    for _ in range(3):
        h.forward(xb, betas, T, out=out, **kw)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    raw.egx_lbs_timing_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); h.forward(xb, betas, T, out=out, **kw); e1.record()
    torch.cuda.synchronize()
    raw.egx_lbs_timing_read(buf, 0)
    items = max(buf[14], 1); stages = max(buf[3], 1); ne = max(buf[12], 1)
    print(f"mode {mode}: launch {e0.elapsed_time(e1):.3f} ms (pose + fused), wave-items {items}, stages/item {stages / items:.1f}")
    print(f"   per wave-item (cycles): total {buf[13] / items:.0f} | GEMM: operand wait {buf[0] / items:.0f}, LDS write + barrier {buf[1] / items:.0f}, "
          f"LDS read + MFMA {buf[2] / items:.0f} | epilogue: skinning {buf[9] / ne:.0f}, sdf brackets + queue {buf[10] / ne:.0f}, flush + counters {buf[11] / ne:.0f}")
