#!/usr/bin/env python3
"""Drop-in for the reference's crowd_ppo/main_crowd_eval.py: 4 humans on a circle of radius 2 m swap places
(main_crowd_eval.py:273-282); every member sees the others' world-space marker boxes as holes of its walkable polygon and
the members act one after the other (DummyCrowdVectorEnv, dummy_vector_env.py:29-128).  Stochastic unless
--deterministic-eval.  Rollouts go to log/eval_results/crowd-4human/crowd4_<member>.pkl (crowd_env_crowd_eval.py:374).

`--num-scenes S` runs S independent 4-human scenes per GPU (BASELINE config 5: "replicas only" - the members of one scene
interact, scenes do not, so there is no collective)."""
import os
import sys

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
    episodes = [[[] for _ in range(S)] for _ in range(G)]
    ep_ret = torch.zeros(G, S, device="cuda")
    done_ret, done_len, done_cnt = [], [], 0
    ep_len = torch.zeros(G, S, device="cuda")
    max_steps = grp.members[0].cfg["max_depth"]
    target_eps = args.test_num
    pol_out = [dict() for _ in range(G)]
    while done_cnt < target_eps:
        for k, m in enumerate(grp.members):
            out = policy(obs[k], out=pol_out[k])
            wpath_before = m.wpath.cpu()
            o, rew, term = m.step(out["act"], auto_reset=False)
            ep_ret[k] += rew
            ep_len[k] += 1
            mb, pp, fr = m.marker_b.cpu(), m.pred_params.cpu(), m.prev_frame.cpu()
            pel = m.joints.reshape(S, 20, -1, 3)[:, :, 0].cpu()
            tm = term.cpu().numpy()
            for s in range(S):
                episodes[k][s].append([mb[s:s + 1], pp[s:s + 1], m.betas[s].cpu(), m.gender, fr[s, :9].reshape(3, 3),
                                       fr[s, 9:].reshape(1, 3), pel[s:s + 1], "2-frame"])
                if tm[s]:
                    save_rollout_results({"wpath": wpath_before[s], "navmesh_path": None, "scene_path": "data/floor.ply"},
                                         episodes[k][s], out_dir, man_id=f"crowd4_{k}" if S == 1 else f"crowd4_s{s}_{k}")
                    episodes[k][s] = []
                    done_ret.append(float(ep_ret[k, s]))
                    done_len.append(float(ep_len[k, s]))
                    done_cnt += 1
                    ep_ret[k, s] = 0
                    ep_len[k, s] = 0
            m.sample_candidates()
            m._injected = True
            m._launch_reset(m.terminated)
            m._injected = False
            obs[k] = m.obs()
    print(f'Final reward: {np.mean(done_ret)}, length: {np.mean(done_len)}')
    stats_path = os.environ.get("EGX_CROWD_STATS")
    if stats_path:   # per-episode returns / lengths (statistical parity of the bf16 policy)
        import json
        with open(stats_path, "w") as f:
            json.dump({"reward": done_ret, "length": done_len, "scenes": S, "humans_per_scene": G}, f)
    return {"rew": float(np.mean(done_ret)), "len": float(np.mean(done_len)), "episodes": done_cnt}


if __name__ == "__main__":
    a = get_args()
    main(a, num_scenes=a.num_scenes or int(os.environ.get("EGX_CROWD_SCENES", "1")))
