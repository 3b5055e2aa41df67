"""Development aid: three `learn()` passes over a filled 4 x 512 rollout without graph replay - the workload behind
profiles/r03_update_experiments.md (run under `rocprofv3 --kernel-trace`, then scripts/trace_sequence.py on the CSV)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tests.test_trainer_gpu import _Args, _filled_batch
from egogen_amd import setup_world as sw
a = _Args(); a.update_graph = False
pol = sw.build_policy(a)
b = _filled_batch(4, 512, 3, pol)
for _ in range(3):
    pol.learn(b, 256, 1)
torch.cuda.synchronize()
