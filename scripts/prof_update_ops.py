#!/usr/bin/env python3
"""Development aid: which ATen ops / kernels one eager PPO minibatch step launches (torch.profiler), to find element-wise
leftovers of the autograd graph."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
from torch.profiler import profile, ProfilerActivity
from egogen_amd import setup_world as sw
from egogen_amd.ppo_policy import RolloutBatch
from crowd_ppo.main_ppo import get_args

args = get_args([])
pol = sw.build_policy(args)
A, n = 512, 4
b = RolloutBatch(n, A, torch.device("cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
for t in (b.state, b.ego, b.dist, b.time, b.act, b.mu, b.logvar, b.logp_old, b.rew, b.values, b.returns, b.adv):
    t.copy_(torch.randn(t.shape, generator=g, device="cuda") * 0.3)
pol._ensure_flat_grads()
pol._flat_optimizer_ready()
idx = torch.randperm(n * A, device="cuda")[:256]
log = torch.zeros(6, device="cuda")
for _ in range(3):
    pol._fwd_bwd(b, idx, None, log); pol._clip_and_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    pol._fwd_bwd(b, idx, None, log); pol._clip_and_step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type.name == "CUDA"]
print("kernels in one minibatch step:", len(ev))
import collections
c = collections.Counter(e.name[:70] for e in ev)
for k, v in c.most_common(40):
    print(f"{v:4d}  {k}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
