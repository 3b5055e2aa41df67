#!/usr/bin/env python3
"""Drop-in for the reference's crowd_ppo/main_crowd_eval.py: 4 humans on a circle of radius 2 m swap places
(main_crowd_eval.py:273-282); every member sees the others' world-space marker boxes as holes of its walkable polygon and
the members act one after the other (DummyCrowdVectorEnv, dummy_vector_env.py:29-128).  Stochastic unless
--deterministic-eval.  Rollouts go to log/eval_results/crowd-4human/crowd4_<member>.pkl (crowd_env_crowd_eval.py:374).

`--num-scenes S` runs S independent 4-human scenes per GPU (BASELINE config 5: "replicas only" - the members of one scene
interact, scenes do not, so there is no collective)."""
import os
import sys

import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from crowd_ppo.main_ppo import get_args  # noqa: E402
from egogen_amd import setup_world as sw, synth  # noqa: E402
from egogen_amd.body_model import BodyModelHandle  # noqa: E402
from egogen_amd.crowd_env import CrowdGroupEnv  # noqa: E402
from egogen_amd.utils import save_rollout_results  # noqa: E402


def main(args, num_scenes=1, num_agents=4, out_dir="log/eval_results/crowd-4human"):
    if not torch.cuda.is_available():
        raise SystemExit("crowd eval needs a HIP device: the MI355X path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    cfg = sw.load_model(box=True)    # main_crowd_eval.py:222-224 -> load_model(box=True) (crowd_ppo/primitive_model.py:74-96)
    bm, _ = sw.load_body_model("male", seed=args.seed, num_verts=args.num_verts)
    body = BodyModelHandle(bm, synth.marker_ids(args.num_verts), synth.feet_vids(args.num_verts))
    prior = sw.build_motion_prior(seed=args.seed, ckpt_dirs=sw.prior_checkpoint_dirs(cfg, "male"))
    vposer = sw.build_vposer(seed=args.seed)
    policy = sw.build_policy(args, policy_cfg=sw.policy_cfg_from_yaml(cfg))
    # BASELINE config 5: the policy's dense layers on the bf16 MFMA (operands rounded to bf16, fp32 accumulate) unless
    # --policy-dtype fp32; everything else (motion prior, SMPL-X, collision) stays fp32
    from egogen_amd import _lib
    dt = getattr(args, "policy_dtype", None) or "bf16"
    _lib.check(_lib.load().egx_policy_set_precision({"fp32": 0, "bf16x2": 2, "bf16": 1}[dt]), "egx_policy_set_precision")
    print("policy dense layers:", {"bf16": "bf16 operands / fp32 accumulate", "bf16x2": "operands as two bf16 terms / fp32 accumulate",
                                   "fp32": "fp32-equivalent (three bf16 terms)"}[dt])
    if args.resume_path:
        policy.load_state_dict(torch.load(args.resume_path, map_location="cuda")["model"])
        print("Loaded agent from: ", args.resume_path)
    policy.eval()
    # start / target pairs on the circle (main_crowd_eval.py:273-282)
    G, S = num_agents, num_scenes
    st = np.zeros((G, S, 2, 3), np.float32)
    for s in range(S):
        t = np.random.rand() + np.arange(G) * (2 * np.pi / G)
        pts = np.zeros((G, 3), np.float32)
        pts[:, 0], pts[:, 1] = 2 * np.cos(t), 2 * np.sin(t)
        for k in range(G):
            st[k, s, 0], st[k, s, 1] = pts[k], pts[(k + G // 2) % G]
    grp = CrowdGroupEnv(S, st, body, prior, vposer, cfg=sw.env_cfg_from_yaml(cfg), seed=args.seed + 100 * local_rank, keep_rollout=True)
    obs = grp.reset()
    ep_ret = torch.zeros(G, S, device="cuda")
    done_ret, done_len, done_cnt = [], [], 0
    ep_len = torch.zeros(G, S, device="cuda")
    target_eps = args.test_num
    pol_out = [dict() for _ in range(G)]
    # Episode bookkeeping stays on the device: every member step appends one snapshot of what a motion primitive of the rollout file
    # holds (crowd_env_crowd_eval.py:177-178 `self.outmps.append`), the termination flags are the ONE host read per member step, and
    # the primitives of the episodes that just ended come over in one copy per field (before: five copies per member step and one
    # per scene, whether or not anything ended).
    hist = [[] for _ in range(G)]                 # hist[k][i] = (marker_b, pred_params, frame, pelvis) of member k's i-th kept step
    t_start = np.zeros((G, S), np.int64)          # first kept step of the running episode of (member, scene)
    # Only the newest HIST_ON_DEVICE steps of a member stay in device memory (24 KB per scene and step: 12 MB per member step at 512
    # scenes); older ones of still-running episodes move to page-locked host memory with asynchronous copies on this stream.
    HIST_ON_DEVICE = 8

    def offload(k):
        n_old = len(hist[k]) - HIST_ON_DEVICE
        for i in range(max(0, n_old)):
            if hist[k][i][0].is_cuda:
                hist[k][i] = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True) for t in hist[k][i])

    def rows(h, f, its, its_host):                # rows `its` of field f of one kept step, wherever it lives
        return h[f][its] if h[f].is_cuda else h[f][its_host].to("cuda", non_blocking=True)
    torch.cuda.synchronize()
    t_loop = time.time()
    while done_cnt < target_eps:
        for k, m in enumerate(grp.members):
            out = policy(obs[k], out=pol_out[k])
            o, rew, term = m.step(out["act"], auto_reset=False)
            ep_ret[k] += rew
            ep_len[k] += 1
            hist[k].append((m.marker_b.clone(), m.pred_params.clone(), m.prev_frame.clone(),
                            m.joints.reshape(S, 20, -1, 3)[:, :, 0].clone()))
            tm = term.cpu().numpy()
            if tm.any():
                idx = np.nonzero(tm)[0]
                it = torch.as_tensor(idx, device="cuda")
                wp, bet = m.wpath[it].cpu(), m.betas[it].cpu()          # the reset below draws new targets: read them first
                rets, lens = ep_ret[k, it].cpu().tolist(), ep_len[k, it].cpu().tolist()
                for t0 in np.unique(t_start[k, idx]):
                    sel = np.nonzero(t_start[k, idx] == t0)[0]         # positions in idx of the episodes that began at step t0
                    its = it[torch.as_tensor(sel, device="cuda")]
                    its_host = torch.as_tensor(idx[sel])
                    steps = hist[k][int(t0):]
                    torch.cuda.current_stream().synchronize()           # offloaded steps: their asynchronous copies have landed
                    mb, pp, fr, pel = (torch.stack([rows(h, f, its, its_host) for h in steps]).cpu() for f in range(4))    # [T, n, ...]
                    for j, pos in enumerate(sel):
                        sc = int(idx[pos])
                        ep = [[mb[t, j:j + 1], pp[t, j:j + 1], bet[pos], m.gender, fr[t, j, :9].reshape(3, 3), fr[t, j, 9:].reshape(1, 3),
                               pel[t, j:j + 1], "2-frame"] for t in range(len(steps))]
                        save_rollout_results({"wpath": wp[pos], "navmesh_path": None, "scene_path": "data/floor.ply"}, ep, out_dir,
                                             man_id=f"crowd4_{k}" if S == 1 else f"crowd4_s{sc}_{k}")
                done_ret += rets
                done_len += lens
                done_cnt += len(idx)
                ep_ret[k, it] = 0
                ep_len[k, it] = 0
                t_start[k, idx] = len(hist[k])
                lo = int(t_start[k].min())
                if lo > 0:                       # nothing before the oldest running episode is needed any more
                    hist[k] = hist[k][lo:]
                    t_start[k] -= lo
            offload(k)
            m.sample_candidates()
            m._injected = True
            m._launch_reset(m.terminated)
            m._injected = False
            obs[k] = m.obs()
    torch.cuda.synchronize()
    t_loop = time.time() - t_loop
    print(f'Final reward: {np.mean(done_ret)}, length: {np.mean(done_len)}')
    stats_path = os.environ.get("EGX_CROWD_STATS")
    if stats_path:   # per-episode returns / lengths (statistical parity of the bf16 policy)
        import json
        with open(stats_path, "w") as f:
            json.dump({"reward": done_ret, "length": done_len, "scenes": S, "humans_per_scene": G}, f)
    return {"rew": float(np.mean(done_ret)), "len": float(np.mean(done_len)), "episodes": done_cnt, "loop_s": t_loop}


if __name__ == "__main__":
    a = get_args()
    main(a, num_scenes=a.num_scenes or int(os.environ.get("EGX_CROWD_SCENES", "1")))
