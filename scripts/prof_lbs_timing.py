#!/usr/bin/env python3
"""Development aid: per-stage cycle breakdown of the bf16x3 blend loop (needs a library built with -DEGX_LBS_TIMING,
path in EGX_LIB)."""
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
out = {}
for _ in range(3):
    h.forward(xb, betas, T, out=out, sdf=scene, R0=R0, T0=T0)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
raw = C.CDLL(os.environ["EGX_LIB"])
raw.egx_lbs_timing_read(buf, 1)
h.forward(xb, betas, T, out=out, sdf=scene, R0=R0, T0=T0)
torch.cuda.synchronize()
raw.egx_lbs_timing_read(buf, 0)
n = buf[3]
print(f"wave-stages {n}: load burst -> data {buf[0]/n:.0f} cyc, LDS write + barrier {buf[1]/n:.0f} cyc, LDS reads + 72 MFMAs {buf[2]/n:.0f} cyc (pure MFMA issue = 2304)")
