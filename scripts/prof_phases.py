#!/usr/bin/env python3
"""Development aid: wall time of collect / process_fn / learn of the bench workload (synchronising between phases)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from types import SimpleNamespace
from egogen_amd import setup_world as sw, synth
from egogen_amd.body_model import BodyModelHandle
from egogen_amd.trainer import Collector

A = 512
bm_np, _ = sw.load_body_model()
body = BodyModelHandle(bm_np, synth.marker_ids(), synth.feet_vids())
prior = sw.build_motion_prior(); vposer = sw.build_vposer()
env = sw.build_env(A, sw.build_scene("single_box", 256), body, prior, vposer, seed=0)
pa = SimpleNamespace(seed=0, lr=3e-4, gamma=0.99, gae_lambda=0.95, max_grad_norm=0.1, vf_coef=1.0, ent_coef=0.01, weight_kld=1.0,
                     rew_norm=False, eps_clip=0.1, value_clip=False, dual_clip=None, norm_adv=True, recompute_adv=False,
                     deterministic_eval=False, update_graph=True)
policy = sw.build_policy(pa); policy.train()
col = Collector(policy, env); col.reset()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t0 = sync(); b = col.collect(4)
    t1 = sync(); policy.process_fn(b)
    t2 = sync(); policy.learn(b, 256, 1)
    t3 = sync()
    if it >= 2: print(f"collect {1e3*(t1-t0):6.2f} ms  process_fn {1e3*(t2-t1):5.2f} ms  learn {1e3*(t3-t2):6.2f} ms", flush=True)
# inside one vector step
obs = col.obs
ev = lambda: torch.cuda.Event(enable_timing=True)
names, marks = [], []
def mark(n): e = ev(); e.record(); names.append(n); marks.append(e)
for rep in range(3):
    names, marks = [], []
    mark("start")
    out = policy(obs); mark("policy")
    env.z.copy_(out["act"])
    env.prior.sample_prior_into(env.state[:, 0], env.state[:, 1], 804, env.betas, env.z, env.Y_gen, env.Yb_gen); mark("sample_prior")
    torch.cuda.synchronize()
    obs, rew, term = env.step(out["act"]); mark("full env.step (incl. prior again)")
    torch.cuda.synchronize()
for i in range(1, len(marks)):
    print(f"{names[i]:36s} {marks[i-1].elapsed_time(marks[i]):7.3f} ms")
