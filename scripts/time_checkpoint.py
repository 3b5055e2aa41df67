"""Development aid (GPU box): what a checkpoint costs the training thread - state_dict + optim_state_dict, the device-to-host
copy, torch.save (profiles/r05_driver_runs.md)."""
import os
import sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from egogen_amd.trainer import CheckpointWriter
from tests.test_trainer_gpu import _Args
from egogen_amd import setup_world as sw
pol = sw.build_policy(_Args())
pol._ensure_flat_grads(); pol._flat_optimizer_ready()
st = lambda: {"model": pol.state_dict(), "optim": pol.optim_state_dict()}
torch.cuda.synchronize()
for i in range(4):
    t0 = time.time(); s = st(); torch.cuda.synchronize(); t1 = time.time(); h = CheckpointWriter._to_host(s); t2 = time.time()
    print(f"state_dict + optim_state_dict {1e3*(t1-t0):.1f} ms, to host {1e3*(t2-t1):.1f} ms")
t0 = time.time(); torch.save(h, "/tmp/x.pth"); print(f"torch.save {1e3*(time.time()-t0):.1f} ms")
