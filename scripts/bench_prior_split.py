#!/usr/bin/env python3
"""Development aid: does the motion prior's latency chain (63 dependent launches of 6 - 9 us) overlap with ITSELF when the
agents are split over two (or four) HIP streams?  One call at A = 512 on one stream against S calls at A = 512 / S on S streams."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import setup_world as sw

A = 512
prior = sw.build_motion_prior(seed=0)
g = torch.Generator().manual_seed(0)
X = (torch.randn(2, A, 201, generator=g) * 0.5).cuda()
betas = torch.randn(A, 10, generator=g).cuda()
z = torch.randn(A, 128, generator=g).cuda()


def run(S, reps=30):
    n = A // S
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    outs = [(torch.empty(18, n, 201, device="cuda"), torch.empty(18, n, 93, device="cuda")) for _ in range(S)]
    xs = [(X[0, k * n:(k + 1) * n].contiguous(), X[1, k * n:(k + 1) * n].contiguous(), betas[k * n:(k + 1) * n].contiguous(),
           z[k * n:(k + 1) * n].contiguous()) for k in range(S)]
    # one prior object per stream: each owns its workspace
    priors = [prior] + [sw.build_motion_prior(seed=0) for _ in range(S - 1)]

    def once():
        main = torch.cuda.current_stream()
        if S == 1:
            priors[0].sample_prior_into(xs[0][0], xs[0][1], 201, xs[0][2], xs[0][3], outs[0][0], outs[0][1])
            return
        for k in range(S):
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                priors[k].sample_prior_into(xs[k][0], xs[k][1], 201, xs[k][2], xs[k][3], outs[k][0], outs[k][1])
        for k in range(S):
            main.wait_stream(streams[k])
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record(); torch.cuda.synchronize()
    t_eager = e0.elapsed_time(e1) / reps * 1e3
    # the same as one replayed HIP graph (parallel branches for S > 1): no host launch cost
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        once()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return t_eager, e0.elapsed_time(e1) / reps * 1e3, outs


t1, g1, o1 = run(1)
print(f"A = {A}, one stream: eager {t1:.1f} us, replayed graph {g1:.1f} us per sample_prior")
for S in (2, 4):
    t, gt, o = run(S)
    Y = torch.cat([a for a, _ in o], dim=1)
    print(f"{S} streams of {A // S} agents: eager {t:.1f} us, replayed graph with {S} branches {gt:.1f} us   "
          f"(max |dY| vs one stream {float((Y - o1[0][0]).abs().max()):.2e})")
