#!/usr/bin/env python3
"""Development aid: does the prior's chain of small kernels overlap with a fused LBS launch on another stream?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import setup_world as sw, synth
from egogen_amd.body_model import BodyModelHandle, SdfScene

A, T = 256, 20
bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
prior = sw.build_motion_prior()
g = torch.Generator().manual_seed(0)
xb = (torch.randn(A * T, 93, generator=g) * 0.2).cuda(); xb[:, 2] += 1
betas = torch.randn(A, 10, generator=g).cuda()
R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
st = (torch.randn(A, 2, 402, generator=g) * 0.3).cuda(); z = torch.randn(A, 128, generator=g).cuda()
Y = torch.empty(18, A, 201, device="cuda"); Yb = torch.empty(18, A, 93, device="cuda")
out = {}
def lbs(): h.forward(xb, betas, T, out=out, sdf=scene, R0=R0, T0=T0)
def pri(): prior.sample_prior_into(st[:, 0], st[:, 1], 804, betas, z, Y, Yb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): lbs()
    with torch.cuda.stream(s2): pri()
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"A={A}: LBS alone {timeit(lbs):.3f} ms, prior alone {timeit(pri):.3f} ms, both on two streams {timeit(both):.3f} ms", flush=True)
