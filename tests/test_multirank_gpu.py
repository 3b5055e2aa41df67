"""GPU tests of the world_size > 1 path (BASELINE configs[3]): two ranks share ONE device and talk over
gloo (RCCL refuses two ranks on one GPU); the code under test - the fused, HIP-graph-replayed PPO update with the
advantage-moment and flat-gradient all-reduces between its two graphs, and bench.py's rank launcher - is the code the
8-GPU run executes over RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Args:
    seed = 0; lr = 3e-4; gamma = 0.99; gae_lambda = 0.95; max_grad_norm = 0.1; vf_coef = 1.0; ent_coef = 0.01
    weight_kld = 0; rew_norm = False; eps_clip = 0.1; value_clip = 0; dual_clip = None; norm_adv = 1; recompute_adv = 0
    deterministic_eval = False


def _fill(b, seed, pol):
    g = torch.Generator().manual_seed(seed)
    r = lambda t, fn: t.copy_(fn(t.shape, generator=g))
    b.state.copy_(torch.randn(b.state.shape, generator=g) * 0.3)
    b.ego.copy_(torch.rand(b.ego.shape, generator=g) * 2 - 1)
    r(b.dist, torch.rand); r(b.time, torch.rand); r(b.act, torch.randn); r(b.adv, torch.randn); r(b.returns, torch.randn)
    noise = 0.05 * torch.randn(b.logp_old.shape, generator=g)
    with torch.no_grad():
        _, mu, sigma = pol._dist_params(b.obs_flat())
        b.logp_old.copy_(pol.log_prob(mu, sigma, b.act.reshape(-1, 128)).reshape(b.logp_old.shape) + noise.to(b.logp_old.device))


def _make_policy(graph):
    from egogen_amd import setup_world as sw
    a = _Args()
    a.update_graph = graph
    return sw.build_policy(a)


def _dp_worker(rank, world, port, n_local, n_steps, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egogen_amd.ppo_policy import RolloutBatch
    pol = _make_policy(True)
    assert pol.world_size == world
    b = RolloutBatch(n_steps, n_local, "cuda")
    _fill(b, 100 + rank, pol)
    pol._perm_gen.manual_seed(7)
    losses = pol.learn(b, n_local * world, 1)                   # n_steps global minibatches of n_local * world rows
    assert not any(v.get("failed") for v in pol._graph_cache.values()), "graph capture fell back to eager"
    assert all(v.get("g2") is not None for v in pol._graph_cache.values()), "world > 1 must replay two graphs per minibatch"
    if rank == 0:
        torch.save({"grad": pol._flat_grad.cpu(), "loss": losses["loss"],
                    "sd": {k: v.cpu() for k, v in pol.state_dict().items()}}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_graph_replayed_update_equals_single_process(tmp_path):
    """GPU twin of tests/test_ppo_cpu.py::test_data_parallel_update_equals_single_process, through the fused custom-autograd
    update replayed as two HIP graphs per minibatch with the all-reduces between them."""
    from egogen_amd.ppo_policy import RolloutBatch
    world, n_local, n_steps = 2, 16, 3
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(world, 29500 + os.getpid() % 2000, n_local, n_steps, out), nprocs=world, join=True)
    dp = torch.load(out)
    # single process, eager, on the data of both ranks: rank r's row i of the shared permutation -> global row
    pol = _make_policy(False)
    parts = []
    for r in range(world):
        b = RolloutBatch(n_steps, n_local, "cuda")
        _fill(b, 100 + r, pol)
        parts.append(b)
    # every rank walks the SAME permutation of its local rows (shared seed), so global minibatch k = rows perm[k-th span] of
    # every rank: lay the ranks' data out so that a single-process permutation reproduces exactly those minibatches
    N = n_steps * n_local
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(7))
    big = RolloutBatch(n_steps, n_local * world, "cuda")

    def flat(b, name):
        t = getattr(b, name)[:b.n]        # observations hold n + 1 time rows (the one after the last step)
        return t.reshape((b.n * b.A,) + tuple(t.shape[2:]))

    order = []
    for k in range(n_steps):                                      # span k of the permutation, rank-major inside the minibatch
        for r in range(world):
            order += [(r, int(i)) for i in perm[k * n_local:(k + 1) * n_local]]
    for name in ("state", "ego", "dist", "time", "act", "adv", "returns", "logp_old"):
        rows = torch.stack([flat(parts[r], name)[i] for r, i in order])
        flat(big, name).copy_(rows)
    # identity "permutation" over the re-ordered rows: minibatch k = rows [k*2n, (k+1)*2n)
    orig_randperm = torch.randperm
    try:
        torch.randperm = lambda n, generator=None: torch.arange(n)
        ref_losses = pol.learn(big, n_local * world, 1)
    finally:
        torch.randperm = orig_randperm
    assert len(ref_losses["loss"]) == len(dp["loss"]) == n_steps
    # logged losses are all-reduced sums of per-rank terms already scaled by 1/n_global = the global minibatch loss
    np.testing.assert_allclose(dp["loss"], ref_losses["loss"], rtol=2e-4, atol=2e-5)
    g_ref, g_dp = pol._flat_grad.cpu(), dp["grad"]                # gradient of the LAST minibatch (after all-reduce + clip)
    assert float(g_ref.abs().max()) > 0
    assert float((g_ref - g_dp).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    moved = sum(int((v.cpu() - dp["sd"][k]).abs().gt(2e-4).sum()) for k, v in pol.state_dict().items())
    assert moved < 5000, moved


def _bench(extra_env, *flags, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--agents", "8", "--batch-size", "8",
           "--num-verts", "1024", "--sdf-res", "32", "--no-cpu-baseline", "--vec-steps", "2"] + list(flags)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_refuses_to_run_on_one_device():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a 1-GPU box")
    r = _bench({}, "--gpus", "2")
    assert r.returncode != 0
    assert "only 1 HIP device" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_bench_gpus2_spawns_two_ranks():
    """bench.py --gpus 2 launches its own ranks; here both sit on one device over gloo (test knobs)."""
    r = _bench({"EGX_SINGLE_DEVICE": "1", "EGX_DIST_BACKEND": "gloo"}, "--gpus", "2")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert res["config"]["agents_per_gpu"] == 4 and res["config"]["agents_total"] == 8
    assert res["config"]["hip_graph_update"] is True
    assert res["allreduce"]["calls_per_step"] == 2 and res["allreduce"]["in_loop_avg_ms"] > 0
    assert res["weak"]["agents_per_gpu"] == 8 and res["weak"]["value"] > 0
    assert res["value"] > 0 and res["roofline"]["avg_launch_ms"] > 0
