"""GPU tests of the world_size > 1 path (BASELINE configs[3]): two ranks share ONE device and talk over
gloo (RCCL refuses two ranks on one GPU); the code under test - the hand-written update chain (`egx_policy_train_step`,
csrc/update3.hip) with global advantage statistics, replayed as two HIP graphs per minibatch with the flat-gradient all-reduce
between them and the weight-image refresh inside the second, and bench.py's rank launcher - is the code the 8-GPU run executes
over RCCL.  Per-rank minibatches are 32 / 64 rows (multiples of 32: otherwise `_train_handle()` returns None and the autograd
nodes run instead - every test here asserts that this did NOT happen)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import free_port

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


def _flat(b, name):
    t = getattr(b, name)[:b.n]            # observations hold n + 1 time rows (the one after the last step)
    return t.reshape((b.n * b.A,) + tuple(t.shape[2:]))


def _probe_obs(n=32):
    g = torch.Generator().manual_seed(555)
    return {"state": (torch.randn(n, 2, 402, generator=g) * 0.3).cuda(), "egosensing": (torch.rand(n, 2, 32, generator=g) * 2 - 1).cuda(),
            "dist": torch.rand(n, generator=g).cuda(), "time": torch.rand(n, generator=g).cuda()}


def _dp_worker(rank, world, port, n_local, n_steps, n_learn, out_path, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      EGX_DP_OVERLAP="1" if overlap else "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egogen_amd.models import PolicyHipRunner
    from egogen_amd.ppo_policy import RolloutBatch
    pol = _make_policy(True)
    assert pol.world_size == world
    b = RolloutBatch(n_steps, n_local, "cuda")
    _fill(b, 100 + rank, pol)
    pol._perm_gen.manual_seed(7)
    losses, grads = [], []
    for _ in range(n_learn):
        losses += pol.learn(b, n_local * world, 1)["loss"]      # n_steps global minibatches of n_local * world rows per pass
        grads.append(pol._flat_grad.cpu().clone())
    # the code under test is the hand-written chain (csrc/update3.hip) replayed as two graphs per minibatch - NOT the autograd
    # fallback, which is what every per-rank minibatch that is not a multiple of 32 rows silently takes
    assert set(pol._train_handles) == {n_local}, pol._train_handles.keys()
    assert pol.update_paths == {"chain+graph": n_steps * n_learn}, pol.update_paths
    assert not any(v.get("failed") for v in pol._graph_cache.values()), "graph capture fell back to eager"
    assert all(v.get("g1") is not None and v.get("g2") is not None and v.get("path") == "chain" for v in pol._graph_cache.values()), \
        "world > 1 must replay two graphs per minibatch around the all-reduce"
    # EGX_DP_OVERLAP=1: the chain in two halves, the actor + critic bucket all-reduced on the communication stream beside the
    # second; default (0): one all-reduce of the whole flat gradient between the two graphs
    assert pol.overlap_allreduce == overlap and len(pol._grad_buckets()) == 2
    assert all((v.get("g1b") is not None) == overlap for v in pol._graph_cache.values())
    # the weight images the next rollout forward reads were re-made by the last replay of the second graph
    probe = _probe_obs()
    o_live = pol._runner.forward(probe)
    o_fresh = PolicyHipRunner(pol.shared_net, pol.actor, pol.critic).forward(probe)   # packs the CURRENT parameters itself
    for k in ("mu", "logvar", "value"):
        assert torch.equal(o_live[k], o_fresh[k]), f"rank {rank}: rollout forward after learn() reads stale weight images ({k})"
    torch.save({"grads": grads, "loss": losses, "sd": {k: v.cpu() for k, v in pol.state_dict().items()},
                "mu": o_live["mu"].cpu(), "value": o_live["value"].cpu()}, out_path + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _single_process_reference(world, n_local, n_steps, n_learn):
    """One process, eager, on the data of both ranks.  Every rank walks the SAME permutation of its local rows (shared seed),
    so global minibatch k of a pass = rows perm[k-th span] of every rank: per pass the ranks' data are laid out so that the
    identity permutation reproduces exactly those minibatches."""
    from egogen_amd.ppo_policy import RolloutBatch
    pol = _make_policy(False)
    parts = []
    for r in range(world):
        b = RolloutBatch(n_steps, n_local, "cuda")
        _fill(b, 100 + r, pol)
        parts.append(b)
    N = n_steps * n_local
    pg = torch.Generator().manual_seed(7)
    losses, grads = [], []
    orig_randperm = torch.randperm
    for _ in range(n_learn):
        perm = orig_randperm(N, generator=pg)
        big = RolloutBatch(n_steps, n_local * world, "cuda")
        order = []
        for k in range(n_steps):                                  # span k of the permutation, rank-major inside the minibatch
            for r in range(world):
                order += [(r, int(i)) for i in perm[k * n_local:(k + 1) * n_local]]
        for name in ("state", "ego", "dist", "time", "act", "adv", "returns", "logp_old"):
            _flat(big, name).copy_(torch.stack([_flat(parts[r], name)[i] for r, i in order]))
        try:
            torch.randperm = lambda n, generator=None: torch.arange(n)
            losses += pol.learn(big, n_local * world, 1)["loss"]
        finally:
            torch.randperm = orig_randperm
        grads.append(pol._flat_grad.cpu().clone())
    assert set(pol._train_handles) == {n_local * world} and pol.update_paths == {"chain": n_steps * n_learn}, pol.update_paths
    return pol, losses, grads


_DP_RUNS = {}   # (world, n_local, n_steps, n_learn, overlap) -> per-rank results: tests that use the same two-rank run share ONE spawn


def _run_dp(tmp_path, world, n_local, n_steps, n_learn, overlap=False):
    key = (world, n_local, n_steps, n_learn, bool(overlap))
    if key not in _DP_RUNS:
        out = str(tmp_path / f"dp{int(overlap)}.pt")
        mp.spawn(_dp_worker, args=(world, free_port(), n_local, n_steps, n_learn, out, overlap),
                 nprocs=world, join=True)
        _DP_RUNS[key] = [torch.load(out + f".{r}") for r in range(world)]
    return _DP_RUNS[key]


def test_bucketed_overlapped_all_reduce_equals_single_all_reduce(tmp_path):
    """Data-parallel update, opt-in overlapped form (EGX_DP_OVERLAP=1) - the chain in two halves, bucket 0 (actor + critic, the
    clipped prefix of the flat gradient) all-reduced on a communication stream while the encoders' backward runs, bucket 1 after
    it, clip + AdamW after both - against the default (ONE all-reduce of the whole flat gradient between the two graphs): the
    same kernels on the same data and an element-wise sum either way.  The two runs agree to the run-to-run noise of the chain
    itself (the loss kernel sums its rows with floating-point atomics: the last bit of the logged loss and of the head gradients
    is not reproducible between two runs of the SAME configuration): losses to 1e-6 relative, the reduced + clipped gradients to
    1e-6 of their norm, the parameters after two passes of three minibatches up to the AdamW sign flips of entries whose gradient
    is at round-off (a handful of 13.2 M); and within each run the two ranks stay bit-identical."""
    world, n_local, n_steps, n_learn = 2, 32, 3, 2
    a = _run_dp(tmp_path, world, n_local, n_steps, n_learn, overlap=True)
    b = _run_dp(tmp_path, world, n_local, n_steps, n_learn, overlap=False)
    for run in (a, b):
        for k, v in run[0]["sd"].items():
            assert torch.equal(v, run[1]["sd"][k]), k
    np.testing.assert_allclose(a[0]["loss"], b[0]["loss"], rtol=1e-6, atol=1e-7)
    for ga, gb in zip(a[0]["grads"], b[0]["grads"]):
        assert float((ga - gb).norm()) <= 1e-6 * float(gb.norm())
    moved = 0
    for k, v in a[0]["sd"].items():
        d = (v - b[0]["sd"][k]).abs()
        assert float(d.max()) <= 2 * 3e-4 * n_steps * n_learn, k
        moved += int((d > 2e-6).sum())
    assert moved <= 2000, moved


def test_chain_halves_write_disjoint_gradient_buckets():
    """What the overlapped all-reduce relies on, checked on the launches themselves: after the HEADS half of the update chain
    (`egx_policy_train_step_heads`) every actor / critic gradient - the bucket [0, n_clip) that is handed to the collective while
    the second half runs - holds its final value, and the ENCODERS half (`egx_policy_train_step_encoders`) writes nothing below
    n_clip and every shared_net gradient above it.  The flat gradient is poisoned with NaN first: an entry a half does not
    write stays NaN; an entry the second half touches in bucket 0 changes its bits."""
    from egogen_amd.ppo_policy import RolloutBatch
    pol = _make_policy(True)
    n = 64
    b = RolloutBatch(1, n, "cuda")
    _fill(b, 5, pol)
    pol.train()
    pol._ensure_flat_grads()
    assert pol._flat_optimizer_ready()
    pol._refresh_images()
    hs = pol._train_handle(n)
    assert hs is not None
    lay = pol._flat_layout()
    n_clip = pol._n_clip
    names = {id(p): k for k, p in pol.state_dict(keep_vars=True).items() if not k.startswith("_actor_critic.")}
    idx = torch.arange(n, device="cuda")
    log = torch.zeros(6, device="cuda")
    pol._flat_grad.fill_(float("nan"))
    pol._fwd_bwd_train_step(hs, b, idx, None, log, part="heads")
    torch.cuda.synchronize()
    after_heads = pol._flat_grad.clone()
    for p_, off, cnt in lay:
        written = bool(torch.isfinite(after_heads[off:off + cnt]).all())
        assert written == (off < n_clip), (names.get(id(p_)), off, n_clip, written)
        assert (off + cnt <= n_clip) or (off >= n_clip), "a tensor straddles the bucket boundary"
    assert any(names[id(p_)].startswith("shared_net.") for p_, off, _ in lay if off >= n_clip)
    assert all(not names[id(p_)].startswith("shared_net.") for p_, off, _ in lay if off < n_clip)
    pol._fwd_bwd_train_step(hs, b, idx, None, log, part="encoders")
    torch.cuda.synchronize()
    after_both = pol._flat_grad.clone()
    assert torch.equal(after_both[:n_clip].view(torch.int32), after_heads[:n_clip].view(torch.int32)), "the encoders half wrote into bucket 0"
    for p_, off, cnt in lay:
        assert bool(torch.isfinite(after_both[off:off + cnt]).all()), names.get(id(p_))
    # and the two halves together are the undivided chain
    pol._flat_grad.fill_(float("nan"))
    pol._fwd_bwd_train_step(hs, b, idx, None, log, part="all")
    torch.cuda.synchronize()
    for p_, off, cnt in lay:
        a, c = pol._flat_grad[off:off + cnt], after_both[off:off + cnt]
        assert float((a - c).abs().max()) <= 1e-5 * max(float(c.abs().max()), 1e-12), names.get(id(p_))


@pytest.mark.parametrize("n_local", [32])   # 32 rows per rank = the 8-way split of the 256-row minibatch (BASELINE configs[3]); 64 rows
# per rank run through the same code in test_bench_gpus2_spawns_two_ranks' weak shape and in the bucket test below (one spawn less)
def test_two_rank_chain_single_step_equals_single_process(tmp_path, n_local):
    """ONE optimiser step on two ranks through the default update path - `egx_policy_train_step` with the global advantage
    statistics (`use_gstats`), graph 1 | all-reduce of the flat gradient | graph 2 (clip AFTER the reduce, AdamW, image refresh):
    the all-reduced + clipped gradient, the logged loss and the parameters equal the single-process step on the concatenated
    minibatch.  Same weights on both sides, so only the summation order of the weight gradients differs."""
    world = 2
    dp = _run_dp(tmp_path, world, n_local, 1, 1)
    pol, ref_losses, ref_grads = _single_process_reference(world, n_local, 1, 1)
    for r in range(world):
        np.testing.assert_allclose(dp[r]["loss"], ref_losses, rtol=2e-5, atol=2e-6)
        g_ref, g_dp = ref_grads[0], dp[r]["grads"][0]
        assert float(g_ref.abs().max()) > 0
        assert float((g_ref - g_dp).abs().max()) <= 1e-5 * float(g_ref.abs().max()), r
        assert float((g_ref - g_dp).norm()) <= 2e-6 * float(g_ref.norm()), r
    for k, v in dp[0]["sd"].items():                              # replicas stay bit-identical
        assert torch.equal(v, dp[1]["sd"][k]), k
    # AdamW's first step moves every parameter by lr * sign(g) whatever |g| is: a gradient entry within round-off of zero may
    # differ by 2 lr between two summation orders; everything else must agree to round-off
    lr = 3e-4
    n_far = 0
    for k, v in pol.state_dict().items():
        d = (v.cpu() - dp[0]["sd"][k]).abs()
        assert float(d.max()) <= 2 * lr * 1.01, (k, float(d.max()))
        n_far += int((d > 2e-6).sum())
    assert n_far <= 200, n_far


def test_two_rank_chain_two_passes_equal_single_process(tmp_path):
    """Two `learn()` passes of three minibatches on two ranks: replays of both graphs with a new index set / new global
    statistics per minibatch, the weight images re-made inside graph 2 feeding the NEXT minibatch's forward and the rollout
    forward after the update."""
    world, n_local, n_steps, n_learn = 2, 32, 3, 2
    dp = _run_dp(tmp_path, world, n_local, n_steps, n_learn)
    pol, ref_losses, ref_grads = _single_process_reference(world, n_local, n_steps, n_learn)
    assert len(ref_losses) == len(dp[0]["loss"]) == n_steps * n_learn
    np.testing.assert_allclose(dp[0]["loss"], ref_losses, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dp[1]["loss"], dp[0]["loss"], rtol=0, atol=0)   # the logged losses are all-reduced: identical
    for p_ in range(n_learn):                                     # gradient of the last minibatch of each pass (after all-reduce + clip)
        g_ref, g_dp = ref_grads[p_], dp[0]["grads"][p_]
        assert float((g_ref - g_dp).norm()) <= 2e-3 * float(g_ref.norm()), p_
    for k, v in dp[0]["sd"].items():
        assert torch.equal(v, dp[1]["sd"][k]), k
    lr, steps = 3e-4, n_steps * n_learn
    moved = 0
    for k, v in pol.state_dict().items():
        d = (v.cpu() - dp[0]["sd"][k]).abs()
        assert float(d.max()) <= 2 * lr * steps, (k, float(d.max()))
        moved += int((d > 2e-5).sum())
    assert moved < 20000, moved                                   # of 13.2 M parameters (Adam sign flips of near-zero gradients)
    # the rollout forward of a rank after the update = the single-process policy's
    o = pol._runner.forward(_probe_obs())
    assert float((o["mu"].cpu() - dp[0]["mu"]).abs().max()) <= 2e-4 and float((o["value"].cpu() - dp[0]["value"]).abs().max()) <= 2e-4


def _rccl_worker(rank, port, n_local, n_steps, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", EGX_FORCE_DP_PATH="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from egogen_amd.ppo_policy import RolloutBatch
    pol = _make_policy(True)
    assert pol.world_size == 1 and pol._dp
    b = RolloutBatch(n_steps, n_local, "cuda")
    _fill(b, 100, pol)
    pol._perm_gen.manual_seed(7)
    losses = []
    for _ in range(2):
        losses += pol.learn(b, n_local, 1)["loss"]
    assert pol.update_paths == {"chain+graph": 2 * n_steps}, pol.update_paths
    assert all(v.get("g1") is not None and v.get("g2") is not None for v in pol._graph_cache.values())
    torch.save({"loss": losses, "sd": {k: v.cpu() for k, v in pol.state_dict().items()}}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_all_reduce_between_graph_replays_single_rank(tmp_path):
    """The collectives of the data-parallel update are RCCL calls issued between two replayed HIP graphs on the same stream
    (moments of the advantages once per pass, the flat 52.7 MB gradient once per minibatch).  Two ranks cannot share one GPU
    under RCCL, so the multi-rank tests above run over gloo; THIS test runs the same code path over RCCL itself with a world of
    one rank (EGX_FORCE_DP_PATH=1): communicator set-up, all-reduce on the compute stream next to captured graphs, tear-down.
    Result = the plain single-process update (a one-rank all-reduce is the identity; the advantage statistics come from the
    float64 moments instead of the fp32 kernel)."""
    n_local, n_steps = 64, 3
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(free_port(), n_local, n_steps, out), nprocs=1, join=True)
    got = torch.load(out)
    from egogen_amd.ppo_policy import RolloutBatch
    pol = _make_policy(False)
    b = RolloutBatch(n_steps, n_local, "cuda")
    _fill(b, 100, pol)
    pol._perm_gen.manual_seed(7)
    ref = []
    for _ in range(2):
        ref += pol.learn(b, n_local, 1)["loss"]
    np.testing.assert_allclose(got["loss"], ref, rtol=2e-4, atol=2e-5)
    moved = 0
    for k, v in pol.state_dict().items():
        d = (v.cpu() - got["sd"][k]).abs()
        assert float(d.max()) <= 2 * 3e-4 * 2 * n_steps, (k, float(d.max()))
        moved += int((d > 2e-5).sum())
    assert moved < 20000, moved


def _bench(extra_env, *flags, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--agents", "32", "--batch-size", "64",
           "--num-verts", "1024", "--sdf-res", "32", "--no-cpu-baseline", "--vec-steps", "2"] + list(flags)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_refuses_to_run_on_one_device():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a 1-GPU box")
    r = _bench({}, "--gpus", "2")
    assert r.returncode != 0
    assert "only 1 HIP device" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_bench_gpus2_spawns_two_ranks(tmp_path):
    """bench.py --gpus 2 launches its own ranks; here both sit on one device over gloo (test knobs).  The stdout line is the
    compact (< 4 KB) record; the full one is the --detail-file sidecar."""
    detail = str(tmp_path / "detail.json")
    r = _bench({"EGX_SINGLE_DEVICE": "1", "EGX_DIST_BACKEND": "gloo"}, "--gpus", "2", "--detail-file", detail)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["roofline"]["frac"] > 0 and line["config"]["workload"]
    assert line["allreduce"]["in_loop_ms_per_step"] > 0 and line["weak"]["value"] > 0 and line["detail"] == "detail.json"
    res = json.load(open(detail))
    assert res["value"] == pytest.approx(line["value"], rel=1e-4)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert res["config"]["agents_per_gpu"] == 16 and res["config"]["agents_total"] == 32
    assert res["config"]["hip_graph_update"] is True
    # 32 rows per rank and minibatch: the hand-written chain, replayed as graphs - never the autograd fallback
    assert set(res["config"]["update_paths"]) == {"chain+graph"}, res["config"]
    # default: ONE all-reduce of the flat gradient per optimiser step between the two graphs (EGX_DP_OVERLAP=1 is opt-in)
    assert res["allreduce"]["buckets"] == 2 and res["allreduce"]["overlapped_with_backward"] is False
    assert res["allreduce"]["calls_per_step"] == 1 and res["allreduce"]["in_loop_avg_ms"] > 0
    assert res["allreduce"]["backend"] == "gloo" and res["allreduce"]["world_size"] == 2 and len(res["allreduce"]["devices"]) == 2
    assert res["allreduce"]["exposed_ms_per_step"] == pytest.approx(res["allreduce"]["in_loop_ms_per_step"])
    assert res["weak"]["agents_per_gpu"] == 32 and res["weak"]["value"] > 0 and set(res["weak"]["update_paths"]) == {"chain+graph"}
    assert res["value"] > 0 and res["roofline"]["avg_launch_ms"] > 0
