#!/usr/bin/env python3
"""Development aid: egx_sample_prior wall time on the GPU for a few agent counts."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import setup_world as sw
prior = sw.build_motion_prior()
for A in (256, 448, 512, 576, 1024):
    g = torch.Generator().manual_seed(0)
    st = (torch.randn(A, 2, 402, generator=g) * 0.3).cuda()
    betas = torch.randn(A, 10, generator=g).cuda(); z = torch.randn(A, 128, generator=g).cuda()
    Y = torch.empty(18, A, 201, device="cuda"); Yb = torch.empty(18, A, 93, device="cuda")
    for _ in range(3):
        prior.sample_prior_into(st[:, 0], st[:, 1], 804, betas, z, Y, Yb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        prior.sample_prior_into(st[:, 0], st[:, 1], 804, betas, z, Y, Yb)
    e1.record(); torch.cuda.synchronize()
    print(f"A={A:5d}  sample_prior {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)
