#!/usr/bin/env python3
"""One configuration of the fused LBS kernel, a few launches (target of rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import synth
from egogen_amd.body_model import BodyModelHandle, SdfScene
A, T = 512, 20
bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
g = torch.Generator().manual_seed(0)
xb = (torch.randn(A * T, 93, generator=g) * 0.2).cuda(); xb[:, 2] += 1
betas = torch.randn(A, 10, generator=g).cuda()
R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
mode = sys.argv[1] if len(sys.argv) > 1 else "sdf"
kw = dict(sdf=scene, R0=R0, T0=T0) if mode == "sdf" else {}
out = {}
for _ in range(4):
    h.forward(xb, betas, T, out=out, **kw)
torch.cuda.synchronize()
