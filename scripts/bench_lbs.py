#!/usr/bin/env python3
"""Micro-benchmark of the SMPL-X/SDF kernels alone (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import synth
from egogen_amd.body_model import BodyModelHandle, SdfScene

bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
for A in (64, 512):
    T = 20
    B = A * T
    g = torch.Generator().manual_seed(0)
    xb = (torch.randn(B, 93, generator=g) * 0.2).cuda(); xb[:, 2] += 1
    betas = torch.randn(A, 10, generator=g).cuda()
    R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
    for name, kw in (("picks", {}), ("picks+sdf", dict(sdf=scene, R0=R0, T0=T0)),
                     ("verts", dict(want_verts=True)), ("verts+sdf", dict(want_verts=True, sdf=scene, R0=R0, T0=T0))):
        out = {}
        for _ in range(3):
            h.forward(xb, betas, T, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            h.forward(xb, betas, T, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        flops = B * 31425 * 469 * 2
        print(f"A={A} B={B} {name:10s} {ms:8.3f} ms  blend {flops/ms/1e9:7.1f} TFLOP/s  "
              f"{'verts %.1f GB/s' % (B*10475*12/ms/1e6) if 'verts' in name else ''}", flush=True)
