"""egogen_amd.trainer.CheckpointWriter: the drivers' `checkpoint_<epoch>.pth` / `policy.pth` (main_ppo.py:186-216) written by a worker
thread.  What must hold: a file holds the state of the moment `save` was called (the caller keeps training), files appear in
submission order and complete (written under a temporary name), a failed write surfaces, `close()` leaves everything on disk."""
import os

import pytest
import torch

from egogen_amd.trainer import CheckpointWriter


def _state(x):
    return {"model": {"w": x}, "optim": {"state": {0: {"step": torch.tensor(3.0), "exp_avg": x * 2}}, "param_groups": [{"lr": 3e-4, "params": [0]}]}}


@pytest.mark.parametrize("sync", [False, True])
def test_snapshots_are_of_the_moment_of_the_call(tmp_path, monkeypatch, sync):
    monkeypatch.setenv("EGX_SYNC_CHECKPOINTS", "1" if sync else "0")
    w = CheckpointWriter(depth=2)
    assert w.sync == sync
    x = torch.zeros(256, 256)
    for i in range(5):
        x.add_(1.0)                                   # the "optimiser" keeps writing the same storage
        w.save(_state(x), str(tmp_path / f"checkpoint_{i}.pth"))
    w.close()
    w.close()                                         # idempotent
    assert sorted(os.listdir(tmp_path)) == [f"checkpoint_{i}.pth" for i in range(5)]      # no *.tmp left behind
    for i in range(5):
        st = torch.load(tmp_path / f"checkpoint_{i}.pth")
        assert float(st["model"]["w"][0, 0]) == i + 1 and float(st["optim"]["state"][0]["exp_avg"][3, 3]) == 2 * (i + 1)
        assert st["optim"]["param_groups"] == [{"lr": 3e-4, "params": [0]}] and st["model"]["w"].device.type == "cpu"


def test_a_failed_write_is_reported(tmp_path, monkeypatch):
    monkeypatch.setenv("EGX_SYNC_CHECKPOINTS", "0")
    w = CheckpointWriter()
    w.save(_state(torch.ones(4)), str(tmp_path / "no_such_dir" / "policy.pth"))
    with pytest.raises(RuntimeError, match="checkpoint write failed"):
        w.flush()
    w.save(_state(torch.ones(4)), str(tmp_path / "policy.pth"))      # the writer stays usable
    w.close()
    assert float(torch.load(tmp_path / "policy.pth")["model"]["w"][0]) == 1.0


@pytest.mark.gpu
def test_device_tensors_are_staged_and_snapshots_stay_intact(tmp_path, monkeypatch):
    """Device tensors go through the page-locked staging buffers (three for depth 2): seven saves reuse each of them, the
    parameters keep changing on the device meanwhile, and every file holds the values of its own moment - also for mixed dtypes,
    non-contiguous and 0-dim tensors and host tensors in the same state."""
    monkeypatch.setenv("EGX_SYNC_CHECKPOINTS", "0")
    w = CheckpointWriter(depth=2)
    p = torch.zeros(1 << 20, device="cuda")
    q = torch.zeros(37, 53, device="cuda").t()                 # non-contiguous
    k = torch.zeros((), device="cuda", dtype=torch.float64)
    idx = torch.arange(5, device="cuda", dtype=torch.int32)
    for i in range(7):
        p.add_(1.0); q.add_(2.0); k.add_(3.0)
        w.save({"model": {"p": p, "q": q, "k": k, "idx": idx}, "optim": {"state": {0: {"step": torch.tensor(float(i))}}, "tag": i}},
               str(tmp_path / f"checkpoint_{i}.pth"))
    w.close()
    for i in range(7):
        st = torch.load(tmp_path / f"checkpoint_{i}.pth")
        m = st["model"]
        assert all(t.device.type == "cpu" for t in m.values())
        assert float(m["p"].min()) == float(m["p"].max()) == i + 1 and m["p"].shape == (1 << 20,)
        assert m["q"].shape == (53, 37) and float(m["q"].min()) == float(m["q"].max()) == 2 * (i + 1)
        assert m["k"].dtype == torch.float64 and float(m["k"]) == 3 * (i + 1) and m["idx"].tolist() == [0, 1, 2, 3, 4]
        assert st["optim"]["tag"] == i and float(st["optim"]["state"][0]["step"]) == i


@pytest.mark.gpu
def test_aliased_entries_are_written_once_and_stay_aliases(tmp_path, monkeypatch):
    """The policy's state_dict names the actor / critic parameters twice (`actor.*` / `critic.*` and `_actor_critic.*`): the
    worker-thread file holds each tensor once - same size as the synchronous torch.save, aliases still sharing storage after
    torch.load - and not the rest of a staging buffer that an earlier, larger state left behind."""
    p = torch.randn(1 << 18, device="cuda")
    big = {"model": {"a": torch.randn(1 << 20, device="cuda")}}
    state = {"model": {"actor.w": p, "_actor_critic.actor.w": p, "other": torch.randn(1 << 10, device="cuda")}}
    monkeypatch.setenv("EGX_SYNC_CHECKPOINTS", "0")
    w = CheckpointWriter(depth=1)
    for i in range(3):                        # every staging slot has held the larger state once
        w.save(big, str(tmp_path / f"big_{i}.pth"))
    w.save(state, str(tmp_path / "async.pth"))
    w.close()
    monkeypatch.setenv("EGX_SYNC_CHECKPOINTS", "1")
    w2 = CheckpointWriter()
    w2.save(state, str(tmp_path / "sync.pth"))
    w2.close()
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    a, b = os.path.getsize(tmp_path / "async.pth"), os.path.getsize(tmp_path / "sync.pth")
    assert abs(a - b) <= 4096 and a < 4 * ((1 << 18) + (1 << 10)) + 65536, (a, b)
    st = torch.load(tmp_path / "async.pth")["model"]
    assert st["actor.w"].data_ptr() == st["_actor_critic.actor.w"].data_ptr()
    assert torch.equal(st["actor.w"], p.cpu()) and torch.equal(st["other"], state["model"]["other"].cpu())
