#!/usr/bin/env python3
"""Development aid: egx_policy_forward wall time on the GPU."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd.models import ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, POLICY_CFG, PolicyHipRunner
torch.manual_seed(0)
from egogen_amd import _lib
_lib.load().egx_policy_set_precision(int(os.environ.get("EGX_POLICY_PREC", "2")))   # 0 f32-equivalent, 2 bf16x2 (drivers' default), 1 bf16
ac = ActorCritic(GAMMAActor(POLICY_CFG), GAMMACritic(POLICY_CFG), GAMMAPolicyBase(POLICY_CFG)).cuda()
run = PolicyHipRunner(ac.shared_net, ac.actor, ac.critic)
for A in (64, 256, 512, 2560):
    g = torch.Generator().manual_seed(0)
    obs = {"state": (torch.randn(A, 2, 402, generator=g) * 0.3).cuda(), "egosensing": torch.rand(A, 2, 32, generator=g).cuda(),
           "dist": torch.rand(A, generator=g).cuda(), "time": torch.rand(A, generator=g).cuda()}
    out = {}
    for _ in range(3):
        run.forward(obs, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run.forward(obs, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"A={A:5d}  policy forward {e0.elapsed_time(e1) / 20:7.3f} ms", flush=True)
