#!/usr/bin/env python3
"""Development probe: the SDF-counting LBS launch on a side stream beside the policy + motion-prior chain on the main stream
(what a split env step would overlap), against the two run back to back."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import setup_world as sw, synth
from egogen_amd.body_model import BodyModelHandle, SdfScene
A, T = 512, 20
bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
prior = sw.build_motion_prior(seed=0)
g = torch.Generator().manual_seed(0)
xb = (torch.randn(A * T, 93, generator=g) * 0.2).cuda(); xb[:, 2] += 1
betas = torch.randn(A, 10, generator=g).cuda()
R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
X = (torch.randn(A, 2, 402, generator=g) * 0.3).cuda()
z = torch.randn(A, 128, generator=g).cuda()
Y = torch.empty(18, A, 201, device="cuda"); Yb = torch.empty(18, A, 93, device="cuda")
out_sdf, out_pk = {}, {}
side = torch.cuda.Stream()
def lbs_sdf(): h.forward(xb, betas, T, out=out_sdf, sdf=scene, R0=R0, T0=T0)
def lbs_picks(): h.forward(xb, betas, T, out=out_pk)
def chain(): prior.sample_prior_into(X[:, 0], X[:, 1], 804, betas, z, Y, Yb)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def serial(): lbs_sdf(); chain()
def split_serial(): lbs_picks(); chain(); lbs_sdf()
def overlapped():
    lbs_picks()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): lbs_sdf()
    chain()
    torch.cuda.current_stream().wait_stream(side)
print(f"WG/CU={os.environ.get('EGX_LBS_WG_PER_CU','2')}: lbs_sdf {timeit(lbs_sdf):.3f}  lbs_picks {timeit(lbs_picks):.3f}  chain {timeit(chain):.3f}  "
      f"serial(lbs_sdf+chain) {timeit(serial):.3f}  picks+chain+sdf {timeit(split_serial):.3f}  overlapped {timeit(overlapped):.3f} ms")
