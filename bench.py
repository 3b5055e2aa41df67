#!/usr/bin/env python3
"""Benchmark of the crowd_ppo hot path: PPO env-steps/sec (BASELINE.json metric).

One "step" = one on-policy cycle over one batch of synthetic input:
    collect  n_vec vector steps of A agents (policy forward + action sampling, C-VAE decode + regressor, SMPL-X
             forward on A*20 bodies with fused SDF counting, VPoser encoder, reward/feature/egosensing kernel,
             auto-reset)                                                          -> n_vec * A transitions
    update   critic values + GAE, then all minibatches of the clipped-PPO update (AdamW, grad-norm clip)
value = transitions of all ranks / wall time of K such steps (max over ranks), inputs resident in HBM.

Workload (BASELINE.json: metric quoted on 512 parallel SMPL-X agents; configs[1]/[2]): A = 512 agents per GPU,
synthetic seeded SMPL-X-shaped body (V = 10475), single-box SDF scene 256^3 by default (`--scene box` = random-box
scene set with the walkability-map penetration term), random-init networks, 4 vector steps per collect (2048
transitions), minibatch 256 per rank -> 8 optimiser steps per collect, repeat 1.

N > 1 (weak scaling): every rank owns A agents and its own scene replica; the only collectives are the
advantage-moment all-reduce (3 doubles) and one flat 52.7 MB gradient all-reduce per optimiser step.

Extra objects on the JSON line:
  roofline      dominant kernel = egx_lbs_fused_kernel (fp32-MFMA blend GEMM + skinning + SDF epilogue);
                achieved = 2*469*31425 FLOP/body * bodies per launch / average launch duration measured with HIP events
                recorded around that kernel inside the timed region; peak = 157.3 TFLOP/s (fp32 MFMA, MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (port of the reference's per-agent, x4-replicated structure) timed on the host cores
                on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_BODY = 2.0 * 469 * 31425          # blend GEMM only: K = 10 betas + 9 x 51 movable joints (jaw/eye columns are exactly 0)
PEAK_F32_MFMA_TFLOPS = 157.3               # dense fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0             # dense bf16 MFMA peak (same guide); the bf16x3 blend issues 6 bf16 products per fp32 product


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--agents", type=int, default=512, help="agents per GPU")
    p.add_argument("--scene", type=str, default="single_box", choices=["single_box", "room0", "box"])
    p.add_argument("--sdf-res", type=int, default=256)
    p.add_argument("--vec-steps", type=int, default=4, help="vector steps per collect")
    p.add_argument("--batch-size", type=int, default=256, help="minibatch per rank")
    p.add_argument("--scaling", type=str, default="weak", choices=["weak", "strong"],
                   help="weak: --agents and --batch-size per GPU (BASELINE configs[3]); strong: both are totals split over the ranks")
    p.add_argument("--num-verts", type=int, default=10475)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-agent-steps", type=int, default=6)
    p.add_argument("--graph", type=int, default=0, help="capture the env step into a HIP graph")
    p.add_argument("--update-graph", type=int, default=1, help="replay the PPO minibatch update as HIP graphs")
    return p.parse_args()


class PolicyArgs:
    seed = 0
    lr = 3e-4
    gamma = 0.99
    gae_lambda = 0.95
    max_grad_norm = 0.1
    vf_coef = 1.0
    ent_coef = 0.01
    weight_kld = 0
    rew_norm = False
    eps_clip = 0.1
    value_clip = 0
    dual_clip = None
    norm_adv = 1
    recompute_adv = 0
    deterministic_eval = False


def cpu_baseline(args, scene, n_agent_steps, budget_s=20.0):
    """Reference-structured CPU path (oracle): one agent at a time, batch replicated x4 (crowd_env_2f.py:29-32), host ray
    casting, plus the CPU cost of the PPO update per transition."""
    from egogen_amd import synth
    from egogen_amd.models import (ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, GAMMAPrimitiveCombo, POLICY_CFG,
                                   PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder)
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    from oracle import nets as onets, ppo as oppo
    cores = min(os.cpu_count() or 1, 32)   # more threads only add fork/join overhead at these tensor sizes
    torch.set_num_threads(cores)
    V = args.num_verts
    bm = synth.make_body_model(0, num_verts=V)
    torch.manual_seed(0)
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    vp = VPoserEncoder().eval()
    psd = {k: v.detach() for k, v in combo.state_dict().items()}
    vsd = {k: v.detach().float() for k, v in vp.state_dict().items()}
    if scene["scene_kind"] == "sdf":
        sd = {k: torch.as_tensor(np.asarray(scene["sdf_dict"][k])) for k in ("sdf", "center", "scale")}
        okw = dict(scene_kind="sdf", sdf_dict=sd, edges=synth.rings_to_edges(scene["rings"]))
        pairs = np.asarray(scene["pairs"][:n_agent_steps], np.float32)
    else:
        okw = dict(scene_kind="box", box_scenes=scene["box_scenes"])
        pairs = np.asarray(scene["box_scenes"][0]["pairs"][:n_agent_steps], np.float32)
    o = OracleCrowdEnv(BodyModel(bm), psd, vsd, synth.marker_ids(V), synth.feet_vids(V), synth.feet_marker_idx(), **okw)
    ms = synth.load_assets()
    rep = 4
    poses = torch.tensor(ms["seed_poses"][5:7, :66], dtype=torch.float32)[None].repeat(rep, 1, 1)
    trans = torch.tensor(ms["seed_trans"][5:7], dtype=torch.float32)[None].repeat(rep, 1, 1)
    betas = torch.tensor(ms["seed_betas"], dtype=torch.float32).reshape(1, 10).repeat(rep, 1)
    ac = ActorCritic(GAMMAActor(POLICY_CFG), GAMMACritic(POLICY_CFG), GAMMAPolicyBase(POLICY_CFG))
    pol_sd = {k: v.detach() for k, v in ac.state_dict().items()}
    t_env = 0.0
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        t_begin = time.perf_counter()
        done = 0
        for i in range(n_agent_steps):
            if done >= 2 and time.perf_counter() - t_begin > budget_s:
                break
            done += 1
            st = torch.as_tensor(pairs[i:i + 1, 0]).repeat(rep, 1)
            tg = torch.as_tensor(pairs[i:i + 1, 1]).repeat(rep, 1)
            t0 = time.perf_counter()
            tr, go, bp, wp = o.next_body(st, tg, poses, trans, betas,
                                         yaw_jitter=None if scene["scene_kind"] == "sdf" else torch.zeros(rep))
            obs, _ = o.reset_from(tr, go, bp, betas, wp, scene_idx=None if scene["scene_kind"] == "sdf" else [0] * rep)
            t_reset = time.perf_counter() - t0
            t0 = time.perf_counter()
            hx = onets.policy_base(pol_sd, obs)
            mu, lv = onets.policy_actor(pol_sd, hx)
            onets.policy_critic(pol_sd, hx)
            z = mu + torch.exp(lv.clamp(-2.5, 2.5)) ** 0.5 * torch.randn(rep, 128, generator=g)
            o.step(z)
            t_env += time.perf_counter() - t0 + t_reset / 11.0   # one reset per ~max_depth steps
    n_agent_steps = done
    per_step = t_env / n_agent_steps
    # PPO update cost per transition: one minibatch of 256 forward+backward+AdamW on the CPU
    ac.train()
    opt = torch.optim.AdamW(ac.parameters(), lr=3e-4, weight_decay=0.01)
    B = 256
    obs = {"state": torch.randn(B, 2, 402), "egosensing": torch.rand(B, 2, 32), "dist": torch.rand(B), "time": torch.rand(B)}
    act, adv, ret, lpo = torch.randn(B, 128), torch.randn(B), torch.randn(B), torch.randn(B) - 180
    t0 = time.perf_counter()
    hx = ac.shared_net(obs)
    (mu, lv), _ = ac.actor(hx)
    loss, _ = oppo.ppo_loss(mu, lv, ac.critic(hx), act, adv, ret, lpo)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(list(ac.actor.parameters()) + list(ac.critic.parameters()), 0.1)
    opt.step()
    per_trans_update = (time.perf_counter() - t0) / B
    return {"value": 1.0 / (per_step + per_trans_update), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_agent_steps} agent-steps of the oracle env (V={V}, batch x4 per agent as in the reference, "
                      f"{per_step * 1e3:.0f} ms each) + one 256-sample PPO minibatch on CPU ({per_trans_update * 1e3:.2f} ms/transition)"}


def _log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    args = get_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # test knobs (never set by the driver): run several ranks on ONE device over gloo to exercise the multi-rank code path
    backend = os.environ.get("EGX_DIST_BACKEND", "nccl")
    if os.environ.get("EGX_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from egogen_amd import _lib, setup_world as sw, synth
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.trainer import Collector
    lib = _lib.load()

    if args.scaling == "strong":  # fixed total work: 512 agents and one 256-sample minibatch over all ranks
        assert args.agents % world == 0 and args.batch_size % world == 0, "--agents / --batch-size must divide by the rank count"
        args.agents //= world
        args.batch_size //= world
    A = args.agents
    pa = PolicyArgs()
    pa.update_graph = bool(args.update_graph)
    bm, _ = sw.load_body_model("male", seed=0, num_verts=args.num_verts)
    body = BodyModelHandle(bm, synth.marker_ids(args.num_verts), synth.feet_vids(args.num_verts))
    prior = sw.build_motion_prior(seed=0)
    vposer = sw.build_vposer(seed=0)
    scene = sw.build_scene(args.scene, sdf_res=args.sdf_res, seed=0)
    _log("assets built")
    env = sw.build_env(A, scene, body, prior, vposer, seed=rank, use_graph=bool(args.graph))
    _log(f"env built ({'%d valid start pairs' % env.valid_pairs.shape[0] if env.valid_pairs is not None else 'box scenes'})")
    policy = sw.build_policy(pa)
    policy.train()
    collector = Collector(policy, env)
    collector.reset()
    n_vec = args.vec_steps
    global_bs = args.batch_size * world

    def one_step():
        batch = collector.collect(n_vec)
        policy.process_fn(batch)
        return policy.learn(batch, global_bs, 1)

    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        _log(f"warmup step {i} done")

    # HIP events around the fused LBS kernel of every vector step in the timed region
    n_ev = args.steps * n_vec
    evs = []
    for _ in range(n_ev):
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(lib.egx_event_create(C.byref(e0)), "event")
        _lib.check(lib.egx_event_create(C.byref(e1)), "event")
        evs.append((e0, e1))
    if not args.graph:
        env.profile_events = list(evs)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.graph:
        # a captured step cannot carry per-launch events: time the same kernel on eager passes right after the region
        env.profile_events = list(evs)
        z = torch.zeros(A, 128, device="cuda")
        for _ in range(n_ev):
            env.z.copy_(z)
            env._step_core()
        torch.cuda.synchronize()
    ms_list = []
    for e0, e1 in evs:
        ms = C.c_float()
        _lib.check(lib.egx_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
        ms_list.append(ms.value)
        lib.egx_event_destroy(e0)
        lib.egx_event_destroy(e1)
    _log(f"timed region done: {elapsed:.3f}s")
    lbs_ms = float(np.mean(ms_list))
    # HBM-side traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass (profiles/r01_lbs_pmc.json);
    # it applies only to the configuration that pass was taken on
    traffic = None
    try:
        blend = int(lib.egx_lbs_get_blend_mode())
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_lbs_pmc_bf16x3.json" if blend == 1 else "r01_lbs_pmc.json")))
        c = pmc["config"]
        if c["agents"] == A and c["num_verts"] == args.num_verts and args.scene == "single_box" and args.sdf_res == 256:
            traffic = pmc["derived"]["hbm_side_bytes_per_launch"]
    except Exception:
        pass
    bodies = A * 20
    achieved = FLOP_PER_BODY * bodies / (lbs_ms * 1e-3) / 1e12
    blend = int(lib.egx_lbs_get_blend_mode())
    if blend == 1:
        # 3-term bf16 split: six bf16 MFMA products per fp32 product -> the matrix-pipe ceiling of the ALGORITHMIC fp32
        # flops is the dense bf16 peak / 6
        kernel_name, peak = "egx_lbs_fused3_kernel", PEAK_BF16_MFMA_TFLOPS / 6.0
        peak_note = "dense bf16 MFMA peak 2500 TFLOP/s / 6 partial products per fp32 product (bf16x3 split, fp32 accumulate)"
        executed = 6 * 2.0 * 480 * (328 * 32 * 3) * bodies / (lbs_ms * 1e-3) / 1e12 if args.num_verts == 10475 else None
    else:
        kernel_name, peak, peak_note, executed = "egx_lbs_fused_kernel", PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA peak", None

    transitions = args.steps * n_vec * A * world
    result = {
        "metric": f"PPO env-steps/sec ({A} parallel SMPL-X agents per GPU)",
        "value": transitions / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32" if blend == 0 else "f32 (blend GEMM operands as 3-term bf16 splits, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"crowd_ppo PPO loop: {A} agents/GPU, scene={args.scene}"
                               f"{'' if args.scene == 'box' else f' SDF {args.sdf_res}^3'}, synthetic SMPL-X body V={args.num_verts}, "
                               f"{n_vec} vector steps/collect ({n_vec * A} transitions/GPU), minibatch {args.batch_size}/GPU, repeat 1",
                   "agents_per_gpu": A, "scene": args.scene, "vec_steps_per_collect": n_vec, "minibatch_per_gpu": args.batch_size,
                   "parallelism": f"dp{world}" if world > 1 else "single", "hip_graph_env": bool(args.graph),
                   "hip_graph_update": bool(args.update_graph) and not any(v.get("failed") for v in policy._graph_cache.values())},
        "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "avg_launch_ms": lbs_ms, "launches": len(ms_list), "bodies_per_launch": bodies,
                     "flop_per_body": FLOP_PER_BODY, "peak_note": peak_note, "executed_bf16_tflops": executed,
                     "frac_of_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _log("cpu baseline ...")
        try:
            result["cpu_baseline"] = cpu_baseline(args, scene, args.cpu_agent_steps)
        except Exception as e:  # the baseline is a report, never a reason to lose the measurement
            result["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
