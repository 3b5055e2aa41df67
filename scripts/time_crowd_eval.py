#!/usr/bin/env python3
"""Development aid: wall time and agent-steps/s of crowd_ppo/main_crowd_eval.py (BASELINE config 5: 4 humans per scene, S scenes
per GPU, stochastic bf16 policy, rollouts written) for several S.  usage: time_crowd_eval.py [S ...]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import importlib
mce = importlib.import_module(os.environ.get("EGX_CROWD_EVAL_MODULE", "crowd_ppo.main_crowd_eval"))
from crowd_ppo.main_ppo import get_args
os.chdir(tempfile.mkdtemp())
for S in [int(a) for a in sys.argv[1:]] or [1, 32, 128]:
    args = get_args([])
    args.test_num = 8 * S
    torch.cuda.synchronize()
    t0 = time.time()
    r = mce.main(args, num_scenes=S)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = r["len"] * r["episodes"]
    print(f"S={S:4d}: {r['episodes']} episodes, mean length {r['len']:.1f}, {dt:.2f} s incl. set-up, loop {r['loop_s']:.2f} s -> {steps / r['loop_s']:.0f} agent-steps/s", flush=True)
