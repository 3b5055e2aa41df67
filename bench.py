#!/usr/bin/env python3
"""Benchmark of the crowd_ppo hot path: PPO env-steps/sec (BASELINE.json metric).

One "step" = one on-policy cycle over one batch of synthetic input:
    collect  n_vec vector steps of A agents (policy forward + action sampling, C-VAE decode + regressor, SMPL-X
             forward on A*20 bodies with fused SDF counting, VPoser encoder, reward/feature/egosensing kernel,
             auto-reset)                                                          -> n_vec * A transitions
    update   critic values + GAE, then all minibatches of the clipped-PPO update (AdamW, grad-norm clip)
value = transitions of all ranks / wall time of K such steps (max over ranks), inputs resident in HBM.

Workload (BASELINE.json: metric quoted on 512 parallel SMPL-X agents; configs[1]/[2]): 512 agents in total, synthetic
seeded SMPL-X-shaped body (V = 10475), single-box SDF scene 256^3 by default (`--scene box` = random-box scene set with
the walkability-map penetration term), random-init networks, 4 vector steps per collect (2048 transitions), global
minibatch 256 -> 8 optimiser steps per collect, repeat 1.

`--gpus N` (N > 1): when not already running under torch.distributed.run, bench.py re-launches itself with N ranks (one
process per GPU, RCCL).  Default `--scaling strong` (SURVEY 8(d) C4 headline): the 512 agents and the 256-sample
minibatch are split over the ranks; `--scaling weak` keeps 512 agents and a 256-sample minibatch PER rank (reported as
the `weak` object of the same JSON line when N > 1).  The only collectives are one float64 all-reduce of the advantage
moments of all minibatches per collect and one flat 52.7 MB gradient all-reduce per optimiser step (`allreduce` object:
time inside the loop, stand-alone time, bus bandwidth).

Extra objects on the JSON line:
  roofline      dominant kernel = fused LBS kernel (blend GEMM + skinning + SDF epilogue); achieved = 2*469*31425
                FLOP/body * bodies per launch / average launch duration measured with HIP events recorded around that
                kernel inside the timed region; peak stated for the blend mode in use; `in_scene` = the same launch timed
                with every body inside the scene (SDF queue path live)
  cpu_baseline  the CPU oracle (port of the reference's per-agent, x4-replicated structure) timed on the host cores on a
                bounded sample (rank 0, N = 1 only): all threads, one thread, and a batched-CPU (A = 64) variant
  other_configs (N = 1) the same loop on BASELINE configs[2] (`--scene box`) and on the reference-default shape
                (256 agents, 1024 transitions per collect)
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_BODY = 2.0 * 469 * 31425          # blend GEMM only: K = 10 betas + 9 x 51 movable joints (jaw/eye columns are exactly 0)
# of which the kernel evaluates the vertex tiles that hold a picked vertex or a vertex of the penetration count (the count
# excludes the feet, crowd_env_2f.py:163-175): `flop_per_body` of the roofline object is 2 * 469 * 3 * that vertex count
BLEND_NAME = {0: "f32", 1: "bf16x3", 2: "bf16x2", 3: "f16mix"}
PEAK_F32_MFMA_TFLOPS = 157.3               # dense fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0             # dense bf16 MFMA peak (same guide)
BLEND_PRODUCTS = {0: None, 1: 6, 2: 3, 3: 34.0 / 30.0}   # 16-bit MFMA products per fp32 product of the split blend modes (3 = f16mix:
#                                             two k-steps of 30 with three bf16 products, 28 with one fp16 product)


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--agents", type=int, default=512, help="agents (total with --scaling strong, per GPU with weak)")
    p.add_argument("--scene", type=str, default="single_box", choices=["single_box", "room0", "box"])
    p.add_argument("--sdf-res", type=int, default=256)
    p.add_argument("--vec-steps", type=int, default=4, help="vector steps per collect")
    p.add_argument("--batch-size", type=int, default=256, help="minibatch (global with --scaling strong, per GPU with weak)")
    p.add_argument("--scaling", type=str, default="strong", choices=["weak", "strong"],
                   help="strong: --agents / --batch-size are totals split over the ranks (512 agents: the BASELINE metric); "
                        "weak: both are per GPU")
    p.add_argument("--also-weak", type=int, default=1, help="N > 1 with --scaling strong: time the weak-scaling shape as well")
    p.add_argument("--num-verts", type=int, default=10475)
    p.add_argument("--skin-weights", type=int, default=4, help="non-zero skinning weights per vertex of the synthetic body (4..16)")
    p.add_argument("--body", type=str, default="iid", choices=["iid", "structured"],
                   help="synthetic body: iid = SURVEY 8(d)'s (i.i.d. noise blend shapes; the headline), structured = blend shapes with "
                        "the structure of a learned model (smooth shape fields, local pose correctives): SDF work items can be culled")
    p.add_argument("--lbs-cull", type=int, default=0, help="free-space culling of SDF work items (opt-in; models whose bound is tight)")
    p.add_argument("--lbs-blend", type=str, default="", choices=["", "f32", "bf16x3", "bf16x2", "f16mix"],
                   help="arithmetic of the LBS blend GEMM (default: the library's default, two bf16 planes)")
    p.add_argument("--update-prec", type=str, default="bf16x2", choices=["f32", "bf16x2", "bf16"],
                   help="arithmetic of the PPO update's products (GAMMAPPOPolicy update_precision): three / two / one bf16 terms per operand")
    p.add_argument("--policy-prec", type=str, default="bf16x2", choices=["f32", "bf16x2", "bf16"],
                   help="arithmetic of the rollout policy's dense layers (egx_policy_set_precision)")
    p.add_argument("--repeats", type=int, default=3, help="timed regions of --steps cycles each; value = the median region")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="time cap of each secondary CPU-baseline leg")
    p.add_argument("--cpu-repeats", type=int, default=2, help="repeats of the headline CPU leg (min / median reported)")
    p.add_argument("--cpu-agent-steps", type=int, default=200, help="agent-steps of each sequential CPU-baseline leg (or the time cap)")
    p.add_argument("--extra-configs", type=int, default=1,
                   help="N = 1: 1 = also time configs[2], the strictly fp32-equivalent arithmetic and the reference-default shape (scalars of "
                        "the stdout line); 2 = secondary shapes as well (detail file only); 0 = none")
    p.add_argument("--detail-file", type=str, default="bench_detail.json", help="full record (sidecar of the <= 4 KB stdout line); '' = none")
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


def _physical_cores():
    try:
        out = subprocess.run(["lscpu", "-p=core,socket"], capture_output=True, text=True, timeout=10).stdout
        return len({ln for ln in out.splitlines() if ln and not ln.startswith("#")}) or None
    except Exception:
        return None


def cpu_baseline(args, scene):
    """Reference-structured CPU path (oracle), SURVEY 8(d): one agent at a time with the batch replicated x4
    (crowd_env_2f.py:29-32), host ray casting, DummyVectorEnv-style sequential loop - timed after warm-up on all host
    threads and on one thread - plus a batched-CPU variant (64 agents in one tensor) so that the gain from batching and
    the gain from the GPU are separable, plus the CPU cost of the PPO update per transition (mean of warm minibatches)."""
    from egogen_amd import synth
    from egogen_amd.models import (ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, GAMMAPrimitiveCombo, POLICY_CFG,
                                   PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder)
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    from oracle import nets as onets, ppo as oppo
    logical = os.cpu_count() or 1
    physical = _physical_cores()
    n_all = min(logical, physical or logical)
    V = args.num_verts
    bm = synth.make_body_model(0, num_verts=V)
    torch.manual_seed(0)
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    vp = VPoserEncoder().eval()
    psd = {k: v.detach() for k, v in combo.state_dict().items()}
    vsd = {k: v.detach().float() for k, v in vp.state_dict().items()}
    sdf_scene = scene["scene_kind"] == "sdf"
    if sdf_scene:
        sd = {k: torch.as_tensor(np.asarray(scene["sdf_dict"][k])) for k in ("sdf", "center", "scale")}
        okw = dict(scene_kind="sdf", sdf_dict=sd, edges=synth.rings_to_edges(scene["rings"]))
        pairs_all = np.asarray(scene["pairs"], np.float32)
    else:
        okw = dict(scene_kind="box", box_scenes=scene["box_scenes"])
        pairs_all = np.asarray(scene["box_scenes"][0]["pairs"], np.float32)
    o = OracleCrowdEnv(BodyModel(bm), psd, vsd, synth.marker_ids(V), synth.feet_vids(V), synth.feet_marker_idx(), **okw)
    ms = synth.load_assets()
    ac = ActorCritic(GAMMAActor(POLICY_CFG), GAMMACritic(POLICY_CFG), GAMMAPolicyBase(POLICY_CFG))
    pol_sd = {k: v.detach() for k, v in ac.state_dict().items()}
    g = torch.Generator().manual_seed(0)

    def seeds(rep):
        return (torch.tensor(ms["seed_poses"][5:7, :66], dtype=torch.float32)[None].repeat(rep, 1, 1),
                torch.tensor(ms["seed_trans"][5:7], dtype=torch.float32)[None].repeat(rep, 1, 1),
                torch.tensor(ms["seed_betas"], dtype=torch.float32).reshape(1, 10).repeat(rep, 1))

    def one_vector_step(pair_rows, rep_each):
        """reset (amortised over ~max_depth steps) + policy + env step for len(pair_rows) * rep_each batch rows."""
        rep = len(pair_rows) * rep_each
        poses, trans, betas = seeds(rep)
        st = torch.as_tensor(pairs_all[pair_rows, 0]).repeat_interleave(rep_each, 0)
        tg = torch.as_tensor(pairs_all[pair_rows, 1]).repeat_interleave(rep_each, 0)
        t0 = time.perf_counter()
        tr, go, bp, wp = o.next_body(st, tg, poses, trans, betas, yaw_jitter=None if sdf_scene else torch.zeros(rep))
        obs, _ = o.reset_from(tr, go, bp, betas, wp, scene_idx=None if sdf_scene else [0] * rep)
        t_reset = time.perf_counter() - t0
        t0 = time.perf_counter()
        hx = onets.policy_base(pol_sd, obs)
        mu, lv = onets.policy_actor(pol_sd, hx)
        onets.policy_critic(pol_sd, hx)
        z = mu + torch.exp(lv.clamp(-2.5, 2.5)) ** 0.5 * torch.randn(rep, 128, generator=g)
        o.step(z)
        return time.perf_counter() - t0 + t_reset / 11.0   # one reset per ~max_depth steps

    def sequential(threads, warm, n_steps, budget_s):
        torch.set_num_threads(threads)
        with torch.no_grad():
            for i in range(warm):
                one_vector_step([i % len(pairs_all)], 4)
            t_env, done, t_begin = 0.0, 0, time.perf_counter()
            while done < n_steps and (done < 3 or time.perf_counter() - t_begin < budget_s):
                t_env += one_vector_step([(warm + done) % len(pairs_all)], 4)
                done += 1
        return t_env / done, done

    def batched(threads, A, budget_s):
        torch.set_num_threads(threads)
        with torch.no_grad():
            one_vector_step(list(range(A)), 1)
            t_env, done, t_begin = 0.0, 0, time.perf_counter()
            while done < 2 or (time.perf_counter() - t_begin < budget_s and done < 20):
                t_env += one_vector_step([(done * A + i) % len(pairs_all) for i in range(A)], 1)
                done += 1
        return t_env / (done * A), done * A

    def update_leg(threads, n_mb=6):
        """one 256-sample minibatch forward + loss + backward + clip + AdamW on the CPU; the first call is warm-up"""
        torch.set_num_threads(threads)
        ac.train()
        opt = torch.optim.AdamW(ac.parameters(), lr=3e-4, weight_decay=0.01)
        B = 256
        ts = []
        for i in range(n_mb):
            gg = torch.Generator().manual_seed(10 + i)
            obs = {"state": torch.randn(B, 2, 402, generator=gg), "egosensing": torch.rand(B, 2, 32, generator=gg),
                   "dist": torch.rand(B, generator=gg), "time": torch.rand(B, generator=gg)}
            act, adv, ret = torch.randn(B, 128, generator=gg), torch.randn(B, generator=gg), torch.randn(B, generator=gg)
            lpo = torch.randn(B, generator=gg) - 180
            t0 = time.perf_counter()
            hx = ac.shared_net(obs)
            (mu, lv), _ = ac.actor(hx)
            loss, _ = oppo.ppo_loss(mu, lv, ac.critic(hx), act, adv, ret, lpo)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(list(ac.actor.parameters()) + list(ac.critic.parameters()), 0.1)
            opt.step()
            ts.append(time.perf_counter() - t0)
        return float(np.mean(ts[1:])) / B, n_mb - 1

    warm = max(2, min(20, args.cpu_agent_steps // 10))
    upd_all, n_upd = update_leg(n_all)
    upd_1, _ = update_leg(1, n_mb=3)
    # secondary legs (time-capped): all physical cores, one thread, 64 agents batched in one tensor
    seq_all, n_seq_all = sequential(n_all, max(1, warm // 4), args.cpu_agent_steps, args.cpu_seconds / 2)
    seq_1, n_seq_1 = sequential(1, max(1, warm // 4), args.cpu_agent_steps, args.cpu_seconds / 2)
    bat_all, n_bat = batched(n_all, 64, args.cpu_seconds / 2)
    # headline leg, to the letter of SURVEY 8(d): 20 warm-up + >= 200 timed agent-steps with NO time cap, on the thread count
    # small-tensor CPU inference likes best (32 on a many-core host: more threads mostly add fork/join overhead), repeated
    # `cpu_repeats` times: min / median of the repeats are reported so that the number is reproducible to a few per cent
    n_head = min(n_all, 32)
    upd_h, _ = update_leg(n_head, n_mb=4) if n_head != n_all else (upd_all, n_upd)
    reps = []
    for r in range(max(1, args.cpu_repeats)):
        t, n_done = sequential(n_head, warm if r == 0 else 2, max(200, args.cpu_agent_steps), 1e9)
        reps.append(t)
    seq_h, n_seq_h = float(np.median(reps)), n_done
    torch.set_num_threads(n_all)
    legs = {
        "sequential_all_threads": {"value": 1.0 / (seq_all + upd_all), "threads": n_all, "agent_steps": n_seq_all, "warmup": warm,
                                   "ms_per_agent_step": seq_all * 1e3, "update_ms_per_transition": upd_all * 1e3},
        "sequential_1_thread": {"value": 1.0 / (seq_1 + upd_1), "threads": 1, "agent_steps": n_seq_1, "warmup": max(1, warm // 4),
                                "ms_per_agent_step": seq_1 * 1e3, "update_ms_per_transition": upd_1 * 1e3},
        "batched_64_all_threads": {"value": 1.0 / (bat_all + upd_all), "threads": n_all, "agent_steps": n_bat, "warmup": 64,
                                   "ms_per_agent_step": bat_all * 1e3, "update_ms_per_transition": upd_all * 1e3},
    }
    head = f"sequential_{n_head}_threads"
    legs[head] = {"value": 1.0 / (seq_h + upd_h), "threads": n_head, "agent_steps": n_seq_h, "warmup": warm,
                  "ms_per_agent_step": seq_h * 1e3, "update_ms_per_transition": upd_h * 1e3, "repeats": len(reps),
                  "value_min": 1.0 / (max(reps) + upd_h), "value_max": 1.0 / (min(reps) + upd_h),
                  "ms_per_agent_step_repeats": [t * 1e3 for t in reps]}
    # the headline is the leg run to spec; it is also the fastest sequential leg on a many-core host (the all-cores leg is
    # several times slower - quoting that one would flatter the GPU)
    return {"value": legs[head]["value"], "unit": "env-steps/s", "cores": n_head, "kind": "port", "headline_leg": head,
            "host": {"os_cpu_count": logical, "lscpu_physical_cores": physical},
            "sample": f"oracle env (V={V}, scene={args.scene}) one agent at a time with the reference's x4-replicated batch: "
                      f"{len(reps)} x {n_seq_h} agent-steps after {warm} warm-up on {n_head} threads (median {seq_h * 1e3:.0f} ms each, "
                      f"min {min(reps) * 1e3:.0f}, max {max(reps) * 1e3:.0f}) + the CPU PPO update ({upd_h * 1e3:.3f} ms/transition, warm "
                      f"256-sample minibatches); secondary legs: all {n_all} cores, one thread, 64 agents batched in one tensor",
            "legs": legs}


def _baseline_config_name(args, world):
    """Which entry of BASELINE.json `configs` a run is."""
    total = args.agents if args.scaling == "strong" else args.agents * world
    if args.scene == "box":
        k = "BASELINE configs[2] (512 agents, random-box scene set, full PPO loop)" if (total == 512 and world == 1) else \
            ("BASELINE configs[3] (512 agents x N env shards, gradient all-reduce) on the configs[2] scene set" if world > 1 else
             f"BASELINE configs[2] scene set at {total} agents")
    elif args.scene == "single_box":
        if world > 1:
            k = "BASELINE configs[3] (N env shards, gradient all-reduce) on the configs[1] scene (single-box SDF)"
        elif total == 512:
            k = ("BASELINE configs[1] scene (static single-box SDF, LBS + SDF kernels) at configs[2]'s scale (512 agents, full PPO "
                 "loop): the heavier hybrid of the two")
        elif total == 64:
            k = "BASELINE configs[1] (64 agents, static single-box SDF scene) inside the full PPO loop"
        else:
            k = f"BASELINE configs[1] scene at {total} agents"
    else:
        k = "BASELINE configs[0] scene (room0-shaped SDF) in the batched loop"
    return k


LINE_BUDGET = 4096     # bytes of the ONE stdout line (the driver keeps a bounded tail of stdout: round 4's 21 KB line did not parse)


def _r(x, nd=4):
    return None if x is None else (round(float(x), nd) if isinstance(x, (float, np.floating)) else x)


def compact_line(full):
    """The ONE JSON line of the bench contract, self-sufficient and below LINE_BUDGET bytes: headline metric, config, the
    dominant kernel's roofline, the CPU baseline and three scalars (BASELINE configs[2] verbatim, the strictly
    fp32-equivalent arithmetic, the reference-default shape).  Everything else of `full` (secondary configurations, in-scene
    launches, CPU legs, notes) goes to bench_detail.json and stderr."""
    rf, cb, cfg = full.get("roofline") or {}, full.get("cpu_baseline"), full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 1), _r(line["ms_per_step"], 4)
    line["config"] = {k: cfg.get(k) for k in ("workload", "agents_total", "agents_per_gpu", "scene", "vec_steps_per_collect",
                                              "minibatch_global", "parallelism", "hip_graph_update", "update_paths") if k in cfg}
    line["roofline"] = {k: _r(rf.get(k), 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                                      "launches", "bodies_per_launch", "flop_per_body", "products_per_fp32_product")}
    # the same launch with every body INSIDE the scene (freshly reset agents): the random-init prior of the timed loop throws most
    # bodies out of the SDF cube, which flatters the in-loop launch for a trained policy
    if (rf.get("in_scene") or {}).get("avg_launch_ms") is not None:
        line["roofline"]["in_scene_avg_launch_ms"] = _r(rf["in_scene"]["avg_launch_ms"], 5)
        line["roofline"]["in_scene_frac"] = _r(rf["in_scene"].get("frac"), 5)
    if rf.get("fixups_last_launch_of_loop") is not None:
        line["roofline"]["fp32_reevaluated_vertices"] = rf["fixups_last_launch_of_loop"]
    if cb is not None:
        line["cpu_baseline"] = {k: _r(cb.get(k), 3) for k in ("value", "unit", "cores", "kind")}
        host = cb.get("host") or {}
        if host.get("lscpu_physical_cores"):     # `cores` = the threads the headline leg used, not the size of the host
            line["cpu_baseline"]["cores_note"] = f"{cb.get('cores')} threads of {host['lscpu_physical_cores']} physical cores (the fastest leg)"
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:200]
    for k in ("value_configs2", "value_fp32_equivalent", "value_reference_shape"):
        line[k] = _r(full.get(k), 1)
    if full.get("regions"):
        line["regions_ms_per_step"] = [_r(t, 4) for t in full["regions"]["ms_per_step"]]
    ar = full.get("allreduce")
    if ar:
        line["allreduce"] = {k: _r(ar.get(k), 4) for k in ("bytes", "backend", "world_size", "nccl_version", "buckets", "overlapped_with_backward",
                                                           "calls_per_step", "in_loop_ms_per_step", "exposed_ms_per_step", "standalone_ms",
                                                           "standalone_busbw_GBps")}
    if full.get("weak"):
        w = full["weak"]
        line["weak"] = {"value": _r(w.get("value"), 1), "ms_per_step": _r(w.get("ms_per_step"), 4), "agents_per_gpu": w.get("agents_per_gpu"),
                        "allreduce_in_loop_ms_per_step": _r((w.get("allreduce") or {}).get("in_loop_ms_per_step"), 4)}
    line["precision"] = {k: v for k, v in (full.get("precision") or {}).items() if k != "note"}
    line["detail"] = full.get("detail_file")
    # never let an optional object or a long string cost the record: drop / shorten until the line fits, always return one
    def fits():
        return len(json.dumps(line)) <= LINE_BUDGET
    for k in ("precision", "weak", "regions_ms_per_step", "allreduce"):
        if fits():
            break
        line.pop(k, None)
    for cut in (120, 60, 0):
        if fits():
            break
        if "cpu_baseline" in line:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:cut]
        line["dtype"] = str(line.get("dtype"))[:max(cut, 16)]
        line["config"]["workload"] = str(line["config"].get("workload"))[:max(2 * cut, 48)]
        line["config"].pop("update_paths", None)
    if not fits():
        line["roofline"] = {k: line["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if not fits():
        line = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "data", "roofline", "detail")}
    return line


def _log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks on this node (one process per GPU)."""
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("EGX_SINGLE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} HIP device(s) visible - refusing to report a {args.gpus}-GPU "
                         f"number from fewer devices (EGX_SINGLE_DEVICE=1 EGX_DIST_BACKEND=gloo runs the ranks on one device, "
                         f"for testing the multi-rank path only)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    _log("launching " + " ".join(cmd))
    sys.exit(subprocess.call(cmd, env=env))


def _measure(args, world, rank, A, batch_local, scene, ops, steps, warmup, with_lbs_events=True, repeats=1):
    """Build env + policy for `A` agents on this rank, run `warmup` untimed and `steps` timed on-policy cycles."""
    from egogen_amd import _lib, setup_world as sw
    from egogen_amd.trainer import Collector
    lib = _lib.load()
    body, prior, vposer = ops
    pa = PolicyArgs()
    pa.update_graph = bool(args.update_graph)
    pa.update_precision = args.update_prec
    env = sw.build_env(A, scene, body, prior, vposer, seed=rank, use_graph=bool(args.graph))
    policy = sw.build_policy(pa)
    policy.train()
    collector = Collector(policy, env)
    collector.reset()
    n_vec = args.vec_steps
    global_bs = batch_local * world
    n_mb = max(1, (n_vec * A) // batch_local)

    def one_step():
        batch = collector.collect(n_vec)
        policy.process_fn(batch)
        return policy.learn(batch, global_bs, 1)

    for i in range(warmup):
        one_step()
        torch.cuda.synchronize()
        _log(f"[A={A}] warmup step {i} done")

    # HIP events around the fused LBS kernel of every vector step in the timed region (recorded by the library on the
    # stream the kernel is launched on)
    evs = []
    if with_lbs_events:
        for _ in range(steps * n_vec):
            e0, e1 = C.c_void_p(), C.c_void_p()
            _lib.check(lib.egx_event_create(C.byref(e0)), "event")
            _lib.check(lib.egx_event_create(C.byref(e1)), "event")
            evs.append((e0, e1))
        if not args.graph:
            env.profile_events = list(evs)
    if world > 1:  # events around every gradient all-reduce of the timed region
        policy.allreduce_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                   for _ in range(2 * (steps * n_mb + 8))]      # two buckets per optimiser step
        policy._allreduce_done = []

    def timed_region():
        """EXACTLY `steps` cycles between barrier + synchronize on both sides; the maximum over the ranks."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed_region()          # the region the LBS launch events and the all-reduce events belong to
    extra_regions = []
    for _ in range(max(0, repeats - 1)):   # the same region again: spread of the measurement (like the CPU leg's repeats)
        env.profile_events = []
        if world > 1:
            policy.allreduce_events = []
        extra_regions.append(timed_region())

    if args.graph and evs:
        # a captured step cannot carry per-launch events: time the same kernel on eager passes right after the region
        env.profile_events = list(evs)
        z = torch.zeros(A, 128, device="cuda")
        for _ in range(len(evs)):
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
    ar = None
    if world > 1:
        in_loop = [a.elapsed_time(b) for a, b in policy._allreduce_done]
        policy.allreduce_events, policy._allreduce_done = [], []
        flat = policy._flat_grad
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        alone_ms = (time.perf_counter() - t1) / 10 * 1e3
        nbytes = flat.numel() * 4
        n_buckets = len(policy._grad_buckets())
        overl = bool(policy.overlap_allreduce and n_buckets == 2 and args.update_graph)
        # overlapped form: the calls alternate bucket 0 (beside the encoders' backward) / bucket 1 (nothing left to hide behind)
        hidden = float(np.sum(in_loop[0::2])) / steps if (overl and in_loop) else 0.0
        exposed = (float(np.sum(in_loop[1::2])) if overl else float(np.sum(in_loop))) / steps if in_loop else None
        try:
            nccl_v = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception:
            nccl_v = None
        devs = [None] * world
        dist.all_gather_object(devs, f"rank {rank}: cuda:{torch.cuda.current_device()} {torch.cuda.get_device_name()}")
        ar = {"bytes": nbytes, "buckets": n_buckets, "bucket_bytes": [int(b.numel()) * 4 for b in policy._grad_buckets()],
              "backend": dist.get_backend(), "world_size": dist.get_world_size(), "nccl_version": nccl_v, "devices": devs,
              "overlapped_with_backward": overl,
              "calls_per_step": n_mb * (n_buckets if overl else 1),
              "in_loop_avg_ms": float(np.mean(in_loop)) if in_loop else None,
              "in_loop_ms_per_step": float(np.sum(in_loop)) / steps if in_loop else None,
              "exposed_ms_per_step": exposed, "beside_backward_ms_per_step": hidden, "standalone_ms": alone_ms,
              "standalone_algbw_GBps": nbytes / (alone_ms * 1e-3) / 1e9,
              "standalone_busbw_GBps": nbytes / (alone_ms * 1e-3) / 1e9 * 2 * (world - 1) / world,
              "note": "in_loop = every all-reduce call of the timed region, events on the issuing stream, including the wait for the slowest "
                      "rank.  Default: ONE call per optimiser step between the two graphs (all of it exposed).  EGX_DP_OVERLAP=1: two "
                      "buckets, bucket 0 beside the encoders' backward (beside_backward: at most ~60 us of it can hide), bucket 1 exposed; "
                      "busbw = algbw * 2(N-1)/N (ring all-reduce)"}
    graphs_ok = bool(args.update_graph) and not any(v.get("failed") for v in policy._graph_cache.values())
    forced = env.forced_accepts() if hasattr(env, "forced_accepts") else 0
    if world > 1:
        t = torch.tensor([forced], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        forced = int(t.item())
    return {"elapsed": elapsed, "regions": [elapsed] + extra_regions, "lbs_ms": ms_list, "env": env, "policy": policy, "allreduce": ar,
            "graphs_ok": graphs_ok, "transitions": steps * n_vec * A * world, "update_paths": dict(policy.update_paths),
            "forced_accepts": forced}


def _lbs_in_scene_ms(env, lib, reps=8):
    """The fused LBS launch with every body INSIDE the scene (freshly reset agents, their two seed frames tiled to 20):
    the bracket table cannot decide vertices next to the floor / obstacles, so the queue path of the SDF epilogue is live
    (the random-init motion prior of the timed loop throws most bodies out of the grid)."""
    from egogen_amd import _lib
    if env.sdf is None:
        return None
    env.reset()
    A = env.A
    xb = env.seed[:, [0, 1] * 10, :].contiguous().reshape(A * 20, 93)
    out = []
    for _ in range(reps):
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(lib.egx_event_create(C.byref(e0)), "event")
        _lib.check(lib.egx_event_create(C.byref(e1)), "event")
        _lib.check(lib.egx_profile_next_lbs(e0, e1), "egx_profile_next_lbs")
        env.bm.forward(xb, env.betas, 20, want_verts=False, sdf=env.sdf, R0=env.R0, T0=env.T0, out=env._lbs_out)
        ms = C.c_float()
        _lib.check(lib.egx_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
        out.append(ms.value)
        lib.egx_event_destroy(e0)
        lib.egx_event_destroy(e1)
    cnt = env._lbs_out["pene_count"].reshape(A, 20)
    inside = float((cnt.sum(1) == 0).float().mean().item())
    return {"avg_launch_ms": float(np.mean(out[1:])), "launches": len(out) - 1, "bodies_per_launch": A * 20,
            "fraction_of_agents_with_zero_penetration": inside, "mean_penetrating_vertices_per_body": float(cnt.float().mean().item())}


def _lbs_penetrating_ms(env, lib, reps=8):
    """As `_lbs_in_scene_ms`, with the agents moved INTO geometry: every freshly reset agent's world frame is translated so
    that its pelvis sits within 0.35 m of the obstacle's centre (scene single_box: the 1 m box at (1.5, 0, 0.5)).  Thousands of
    vertices per body are then inside the obstacle or within its level-set band: the bracket table decides the clear cases,
    the undecided ones take the queue + eight-corner path of the SDF epilogue - the launch a policy that walks into obstacles
    would produce.  (State is restored by the next reset.)"""
    from egogen_amd import _lib
    if env.sdf is None:
        return None
    env.reset()
    A = env.A
    g = torch.Generator(device=env.T0.device).manual_seed(1)
    T0 = env.T0.clone()
    T0[:, 0] = 1.5 + (torch.rand(A, generator=g, device=T0.device) - 0.5) * 0.7
    T0[:, 1] = 0.0 + (torch.rand(A, generator=g, device=T0.device) - 0.5) * 0.7
    xb = env.seed[:, [0, 1] * 10, :].contiguous().reshape(A * 20, 93)
    out = []
    for _ in range(reps):
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(lib.egx_event_create(C.byref(e0)), "event")
        _lib.check(lib.egx_event_create(C.byref(e1)), "event")
        _lib.check(lib.egx_profile_next_lbs(e0, e1), "egx_profile_next_lbs")
        env.bm.forward(xb, env.betas, 20, want_verts=False, sdf=env.sdf, R0=env.R0, T0=T0, out=env._lbs_out)
        ms = C.c_float()
        _lib.check(lib.egx_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
        out.append(ms.value)
        lib.egx_event_destroy(e0)
        lib.egx_event_destroy(e1)
    cnt = env._lbs_out["pene_count"].reshape(A, 20)
    return {"avg_launch_ms": float(np.mean(out[1:])), "launches": len(out) - 1, "bodies_per_launch": A * 20,
            "fraction_of_bodies_with_penetration": float((cnt > 0).float().mean().item()),
            "mean_penetrating_vertices_per_body": float(cnt.float().mean().item())}


def main():
    """One JSON line on stdout whatever happens: the measurement, or - when any rank fails - a line with `value` null and
    `error` naming the rank and the exception (the traceback still goes to stderr and the exit status stays non-zero)."""
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:
        import traceback
        traceback.print_exc()
        rank, world = os.environ.get("RANK", "0"), int(os.environ.get("WORLD_SIZE", "1"))
        print(json.dumps({"metric": "PPO env-steps/sec (parallel SMPL-X agents)", "value": None, "unit": "env-steps/s", "n_gpus": world,
                          "higher_is_better": True, "data": "synthetic",
                          "error": f"rank {rank} of {world}: {type(e).__name__}: {str(e)[:600]}",
                          "env": {k: os.environ.get(k) for k in ("EGX_DP_OVERLAP", "EGX_DIST_BACKEND", "EGX_LBS_BLEND", "LOCAL_RANK")}}), flush=True)
        os._exit(1)   # do not wait in a collective's destructor for ranks that are already gone


def _main():
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(600, repeat=True, file=sys.stderr)
    args = get_args()
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the two must agree")
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
    lib = _lib.load()

    if args.scaling == "strong":  # fixed total work: 512 agents and one 256-sample minibatch over all ranks
        assert args.agents % world == 0 and args.batch_size % world == 0, "--agents / --batch-size must divide by the rank count"
        A, bs_local = args.agents // world, args.batch_size // world
    else:
        A, bs_local = args.agents, args.batch_size
    _lib.check(lib.egx_lbs_set_culling(int(args.lbs_cull)), "egx_lbs_set_culling")
    if args.skin_weights != 4 or args.body == "structured":
        bm = synth.make_body_model(0, num_verts=args.num_verts, nnz_weights=args.skin_weights, structured=args.body == "structured")
    else:
        bm, _ = sw.load_body_model("male", seed=0, num_verts=args.num_verts)
    body = BodyModelHandle(bm, synth.marker_ids(args.num_verts), synth.feet_vids(args.num_verts))
    if args.lbs_blend:
        _lib.check(lib.egx_lbs_set_blend_mode({"f32": 0, "bf16x3": 1, "bf16x2": 2, "f16mix": 3}[args.lbs_blend]), "egx_lbs_set_blend_mode")
    _lib.check(lib.egx_policy_set_precision({"f32": 0, "bf16x2": 2, "bf16": 1}[args.policy_prec]), "egx_policy_set_precision")
    ops = (body, sw.build_motion_prior(seed=0), sw.build_vposer(seed=0))
    scene = sw.build_scene(args.scene, sdf_res=args.sdf_res, seed=0)
    _log("assets built")
    m = _measure(args, world, rank, A, bs_local, scene, ops, args.steps, args.warmup, repeats=args.repeats)
    ms_list = m["lbs_ms"]
    regions = m["regions"]
    elapsed = float(np.median(regions))      # `value` / `ms_per_step` = the MEDIAN of the timed regions (each exactly --steps cycles)
    _log("timed regions done: " + ", ".join(f"{t:.3f}s" for t in regions))
    assert m["forced_accepts"] == 0, f"{m['forced_accepts']} episodes started in penetration (no valid start among the reset draws)"
    lbs_ms = float(np.mean(ms_list))
    blend = int(lib.egx_lbs_get_blend_mode())
    # HBM-side traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass (profiles/*_lbs_pmc*.json); it
    # applies only to the configuration that pass was taken on
    traffic = None
    try:
        for name in (f"r06_lbs_pmc_mode{blend}.json", f"r05_lbs_pmc_mode{blend}.json", f"r04_lbs_pmc_mode{blend}.json", f"r03_lbs_pmc_mode{blend}.json",
                     f"r02_lbs_pmc_mode{blend}.json"):   # newest round first
            f = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(f):
                continue
            pmc = json.load(open(f))
            c = pmc["config"]
            if c["agents"] == A and c["num_verts"] == args.num_verts and args.scene == "single_box" and args.sdf_res == 256:
                traffic = pmc["derived"]["hbm_side_bytes_per_launch"]
            break
    except Exception:
        pass
    bodies = A * 20
    bm_handle = m["env"].bm
    verts_eval = bm_handle.lbs_vertices["sdf" if m["env"].sdf is not None else "picks"]
    FLOP_PER_BODY = 2.0 * 469 * 3 * verts_eval     # noqa: N806 - the work this call form needs (see the module constant)
    achieved = FLOP_PER_BODY * bodies / (lbs_ms * 1e-3) / 1e12
    if blend in (1, 2, 3):
        # n-term bf16 split: BLEND_PRODUCTS bf16 MFMA products per fp32 product -> the matrix-pipe ceiling of the
        # ALGORITHMIC fp32 flops is the dense bf16 peak / that count
        npr = BLEND_PRODUCTS[blend]
        if blend == 3 and m["env"].sdf is not None:
            # the tiles that hold picked vertices run the two-plane product (3 per fp32 product), the count-only tiles 34/30
            vp = bm_handle.lbs_vertices["picks"]
            npr = (3.0 * vp + (34.0 / 30.0) * (verts_eval - vp)) / verts_eval
        elif blend == 3:
            npr = 3.0   # a call without counts evaluates the picked tiles only
        kernel_name, peak = "egx_lbs_fused3_kernel", PEAK_BF16_MFMA_TFLOPS / npr
        peak_note = (f"dense 16-bit MFMA peak 2500 TFLOP/s / {npr:.3g} products per fp32 product "
                     f"({BLEND_NAME[blend]}, fp32 accumulate)")
        executed = npr * 2.0 * 480 * (328 * 32 * 3) * bodies / (lbs_ms * 1e-3) / 1e12 if args.num_verts == 10475 else None   # all 328 tiles: upper bound
    else:
        kernel_name, peak, peak_note, executed = "egx_lbs_fused_kernel", PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA peak", None
    def _cull_stats():
        if not (bm_handle.culls and args.lbs_cull and m["env"].sdf is not None and blend in (1, 2)):
            return None
        act, tot = bm_handle.cull_stats(bodies)
        return {"items_evaluated": act, "items_total": tot, "fraction_evaluated": act / max(1, tot)}
    cull_loop = _cull_stats()          # the last launch of the timed loop
    # mixed blend: vertices the last launch re-evaluated in fp32 (the ones its fp16 product / two-plane skinning could not decide)
    fix_loop = bm_handle.fix_stats(bodies) if (blend == 3 and m["env"].sdf is not None) else None
    in_scene = _lbs_in_scene_ms(m["env"], lib)
    if in_scene is not None:
        in_scene["culling"] = _cull_stats()
        in_scene["fixups"] = bm_handle.fix_stats(bodies) if blend == 3 else None
    penetrating = _lbs_penetrating_ms(m["env"], lib) if args.scene == "single_box" else None
    if penetrating is not None:
        penetrating["culling"] = _cull_stats()
        penetrating["fixups"] = bm_handle.fix_stats(bodies) if blend == 3 else None
    if penetrating is not None:
        penetrating["achieved"] = FLOP_PER_BODY * bodies / (penetrating["avg_launch_ms"] * 1e-3) / 1e12
        penetrating["frac"] = penetrating["achieved"] / peak
    other_mode = None
    if in_scene is not None:
        in_scene["achieved"] = FLOP_PER_BODY * bodies / (in_scene["avg_launch_ms"] * 1e-3) / 1e12
        in_scene["frac"] = in_scene["achieved"] / peak
        if blend in (1, 2, 3):  # the same launch in another split mode (three planes <-> two planes), for the record
            alt = 3 - blend if blend in (1, 2) else 2
            _lib.check(lib.egx_lbs_set_blend_mode(alt), "egx_lbs_set_blend_mode")
            try:
                o = _lbs_in_scene_ms(m["env"], lib)
            finally:
                _lib.check(lib.egx_lbs_set_blend_mode(blend), "egx_lbs_set_blend_mode")
            o_peak = PEAK_BF16_MFMA_TFLOPS / BLEND_PRODUCTS[alt]
            o["achieved"] = FLOP_PER_BODY * bodies / (o["avg_launch_ms"] * 1e-3) / 1e12
            other_mode = {"blend": BLEND_NAME[alt], "in_scene_avg_launch_ms": o["avg_launch_ms"], "achieved": o["achieved"],
                          "peak": o_peak, "frac": o["achieved"] / o_peak}

    total_agents = A * world
    result = {
        "metric": f"PPO env-steps/sec ({total_agents} parallel SMPL-X agents)",
        "value": m["transitions"] / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "regions": {"n": len(regions), "steps_each": args.steps, "ms_per_step": [t / args.steps * 1e3 for t in regions],
                    "value_min": m["transitions"] / max(regions), "value_median": m["transitions"] / elapsed,
                    "value_max": m["transitions"] / min(regions)},
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": ("f32" if blend == 0 else ("f32 (blend GEMM: shape/template columns as 2-term bf16 splits, pose correctives as one fp16 product, "
                                            "fp32 accumulate)" if blend == 3 else f"f32 (blend GEMM operands as {3 if blend == 1 else 2}-term bf16 splits, fp32 accumulate)"))
                 + {"f32": "", "bf16x2": "; PPO update products on 2-term bf16 splits", "bf16": "; PPO update products on bf16 operands"}[args.update_prec]
                 + {"f32": "", "bf16x2": "; rollout policy layers on 2-term bf16 splits", "bf16": "; rollout policy layers on bf16 operands"}[args.policy_prec],
        "precision": {"lbs_blend": BLEND_NAME[blend], "ppo_update": args.update_prec, "rollout_policy": args.policy_prec,
                      "motion_prior": "f32 (3-term bf16 splits, 2^-24)", "accumulate": "f32",
                      "note": "bf16xN = every fp32 operand carried as N bf16 terms (8 N significant bits), partial products on the bf16 "
                              "MFMA, fp32 accumulation; the strictly fp32-equivalent configuration (--lbs-blend bf16x3 --update-prec f32 "
                              "--policy-prec f32) is timed in other_configs"},
        "data": "synthetic",
        "config": {"workload": f"{_baseline_config_name(args, world)} - crowd_ppo PPO loop: {total_agents} agents over {world} GPU(s) ({A}/GPU), scene={args.scene}"
                               f"{'' if args.scene == 'box' else f' SDF {args.sdf_res}^3'}, synthetic SMPL-X body V={args.num_verts}, "
                               f"{args.vec_steps} vector steps/collect ({args.vec_steps * total_agents} transitions), "
                               f"global minibatch {bs_local * world}, repeat 1",
                   "agents_total": total_agents, "agents_per_gpu": A, "scene": args.scene, "vec_steps_per_collect": args.vec_steps,
                   "minibatch_global": bs_local * world, "minibatch_per_gpu": bs_local,
                   "parallelism": f"dp{world}" if world > 1 else "single", "hip_graph_env": bool(args.graph),
                   "hip_graph_update": m["graphs_ok"], "update": "hand-written launch chain (csrc/update3.hip)" if m["policy"]._train_handles else "autograd nodes + library GEMMs",
                   "update_paths": m["update_paths"]},
        "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "avg_launch_ms": lbs_ms, "launches": len(ms_list), "bodies_per_launch": bodies,
                     "flop_per_body": FLOP_PER_BODY, "products_per_fp32_product": npr if blend in (1, 2, 3) else None, "vertices_evaluated": verts_eval, "vertices_total": bm_handle.V,
                     "peak_note": peak_note, "executed_bf16_tflops": executed,
                     "frac_of_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS, "in_scene": in_scene, "in_scene_penetrating": penetrating,
                     "culling": {"model_allows": bool(bm_handle.culls), "reference_margin_m": bm_handle.cull_reference_margin,
                                 "enabled": bool(args.lbs_cull), "last_launch_of_loop": cull_loop},
                     "fixups_last_launch_of_loop": fix_loop, "other_blend_mode": other_mode,
                     "sustained_matrix_rate_note": "72 back-to-back v_mfma_f32_32x32x16_bf16 take 46, not 32, cycles each on this part "
                                                   "(clock-limited: 1720 of 2500 TFLOP/s in a load-free micro-benchmark, profiles/r01_ubench.md "
                                                   "section 4, profiles/r02_lbs_experiments.md); `peak` is the data-sheet figure"},
    }
    if world > 1:
        result["allreduce"] = m["allreduce"]
    del m
    torch.cuda.empty_cache()
    if world > 1 and args.scaling == "strong" and args.also_weak:
        mw = _measure(args, world, rank, args.agents, args.batch_size, scene, ops, args.steps, args.warmup, with_lbs_events=False)
        result["weak"] = {"value": mw["transitions"] / mw["elapsed"], "unit": "env-steps/s", "ms_per_step": mw["elapsed"] / args.steps * 1e3,
                          "agents_per_gpu": args.agents, "minibatch_per_gpu": args.batch_size, "allreduce": mw["allreduce"],
                          "update_paths": mw["update_paths"]}
        del mw
        torch.cuda.empty_cache()
    if world == 1 and args.extra_configs and args.scene == "single_box" and args.agents == 512:
        # each extra configuration in a fresh process (a second environment + policy inside this one measurably disturbs the
        # timing: allocator state, captured graphs of the first policy).  --extra-configs 1 (default): the three whose values
        # are scalars of the stdout line; 2: the secondary ones as well (detail file only)
        first = (("value_configs2", "BASELINE configs[2]: 512 agents, random-box scene set (walkability-map penetration term)", ["--scene", "box"]),
                 ("value_fp32_equivalent", "headline workload, STRICTLY fp32-equivalent arithmetic everywhere (LBS blend, PPO update and rollout "
                  "policy on three-term bf16 splits: 2^-24 per product)", ["--lbs-blend", "bf16x3", "--update-prec", "f32", "--policy-prec", "f32"]),
                 ("value_reference_shape", "reference default shape: 256 agents, 1024 transitions per collect, single-box SDF scene", ["--agents", "256"]))
        second = ((None, "per-rank shape of the 4-way strong split (configs[3] at N = 4): 128 agents, 64-sample minibatch",
                   ["--agents", "128", "--batch-size", "64"]),
                  (None, "per-rank shape of the 8-way strong split (configs[3] at N = 8): 64 agents, 32-sample minibatch",
                   ["--agents", "64", "--batch-size", "32"]),
                  (None, "headline workload with the policy's dense layers (rollout + update) on bf16 operands: north_star's 'bf16 MFMA'",
                   ["--update-prec", "bf16", "--policy-prec", "bf16"]),
                  (None, "headline workload on a synthetic body with 12 skinning weights per vertex (real SMPL-X has 4..~12)", ["--skin-weights", "12"]))
        others = []
        for key, label, flags in first + (second if args.extra_configs >= 2 else ()):
            cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--extra-configs", "0", "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--num-verts", str(args.num_verts), "--sdf-res", str(args.sdf_res),
                   "--vec-steps", str(args.vec_steps), "--batch-size", str(args.batch_size), "--detail-file", ""] + flags
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
                r2 = json.loads(line)
                others.append({"workload": label, "value": r2["value"], "unit": r2["unit"], "ms_per_step": r2["ms_per_step"],
                               "regions_ms_per_step": r2.get("regions_ms_per_step"), "lbs_avg_launch_ms": r2["roofline"]["avg_launch_ms"],
                               "lbs_frac": r2["roofline"]["frac"], "lbs_peak": r2["roofline"]["peak"], "steps": r2["steps"],
                               "precision": r2.get("precision"), "config": r2.get("config"), "dtype": r2.get("dtype"),
                               "command": " ".join(cmd[1:])})
                if key:
                    result[key] = r2["value"]
                _log(f"extra config done: {label[:60]}: {r2['value']:.0f} env-steps/s")
            except Exception as e:
                others.append({"workload": label, "value": None, "error": f"{type(e).__name__}: {e}"})
                _log(f"extra config FAILED: {label[:60]}: {type(e).__name__}: {e}")
        result["other_configs"] = others
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _log("cpu baseline ...")
        try:
            result["cpu_baseline"] = cpu_baseline(args, scene)
        except Exception as e:  # the baseline is a report, never a reason to lose the measurement
            result["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        detail = args.detail_file
        result["detail_file"] = os.path.basename(detail) if detail else None
        if detail:      # the full record (secondary configurations, in-scene launches, CPU legs, notes): a sidecar, never stdout
            try:
                with open(detail if os.path.isabs(detail) else os.path.join(ROOT, detail), "w") as f:
                    json.dump(result, f, indent=1)
            except OSError as e:
                _log(f"could not write {detail}: {e}")
            for k in ("roofline", "cpu_baseline", "other_configs"):
                if result.get(k) is not None:
                    _log(f"detail {k}: " + json.dumps(result[k]))
        print(json.dumps(compact_line(result)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
