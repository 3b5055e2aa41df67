#!/usr/bin/env python3
"""Drop-in for the reference's crowd_ppo/main_egobody_eval.py: two people walk towards each other's start in an EgoBody scene
(`exp_data/<scene-name>/navmesh_tight.ply`), one episode of exactly max_depth primitives each; the rollouts go to
./egobody_tmp_res/<member>.pkl (crowd_env_egobody_eval.py:385) for experiments/gen_egobody_{depth,rgb}.py to render
(`egogen_amd.utils.rollout_primitives` is their roll-out step).

Where the reference draws ONE pair per process and exits on its two data filters (pelvis outside the walkable polygon during
the first 5 steps, mean VPoser norm > 14; crowd_env_egobody_eval.py:208-216,229-234), `--num-scenes S` runs S independent
pairs per GPU - male pairs and female pairs as two groups with their own body model and motion prior
(Egobody.next_body, environments.py:780) - and sequences that trip a filter are dropped instead of written.
Without `exp_data/<scene-name>` the in-tree Replica room_0 navmesh stands in as the scene."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from crowd_ppo.main_ppo import get_args  # noqa: E402
from egogen_amd import setup_world as sw, synth  # noqa: E402
from egogen_amd.body_model import BodyModelHandle  # noqa: E402
from egogen_amd.crowd_env import CrowdGroupEnv  # noqa: E402
from egogen_amd.egobody import EgobodySampler  # noqa: E402
from egogen_amd.utils import save_rollout_results  # noqa: E402


def load_motion_seeds(seed_dir="data/locomotion"):
    """main_egobody_eval.py:223-224: every *.npz of data/locomotion; the packaged subseq_00343 when the directory is absent."""
    seeds = []
    if os.path.isdir(seed_dir):
        for fn in sorted(os.listdir(seed_dir)):
            if fn.endswith(".npz"):
                d = np.load(os.path.join(seed_dir, fn))
                seeds.append({"poses": d["poses"], "trans": d["trans"]})
    if not seeds:
        a = synth.load_assets()
        seeds.append({"poses": a["seed_poses"], "trans": a["seed_trans"]})
    return seeds


def main(args, num_scenes=1, scene_name="cab_e", out_dir="./egobody_tmp_res/", static_scene=True, keep_invalid=False):
    if not torch.cuda.is_available():
        raise SystemExit("EgoBody eval needs a HIP device: the MI355X path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    cfg = sw.load_model(box=True)                       # main_egobody_eval.py:213-215
    scene_dir = os.path.join("exp_data", scene_name)
    seeds = load_motion_seeds()
    if os.path.exists(os.path.join(scene_dir, "navmesh_tight.ply")):
        sampler = EgobodySampler.from_scene_dir(scene_dir, seeds, seed=args.seed + 100 * local_rank)
    else:
        a = synth.load_assets()
        print(f"{scene_dir}/navmesh_tight.ply not found: using the packaged Replica room_0 navmesh")
        sampler = EgobodySampler(a["room0_nav_v"], a["room0_nav_f"], seeds, seed=args.seed + 100 * local_rank,
                                 scene_path="data/room_0/mesh_floor.ply", navmesh_path="data/room_0/navmesh_tight.ply")
    vposer = sw.build_vposer(seed=args.seed)
    policy = sw.build_policy(args, policy_cfg=sw.policy_cfg_from_yaml(cfg))
    if args.resume_path:
        policy.load_state_dict(torch.load(args.resume_path, map_location="cuda")["model"])
        print("Loaded agent from: ", args.resume_path)
    policy.eval()
    pairs = [sampler.next_body() for _ in range(num_scenes)]
    results = {"written": [], "dropped": [], "rew": [], "len": []}
    for gi, gender in enumerate(("male", "female")):
        idx = [s for s, p in enumerate(pairs) if p[0]["gender"] == gender]
        if not idx:
            continue
        S = len(idx)
        bm, _ = sw.load_body_model(gender, seed=args.seed + gi, num_verts=args.num_verts)
        body = BodyModelHandle(bm, synth.marker_ids(args.num_verts), synth.feet_vids(args.num_verts))
        prior = sw.build_motion_prior(seed=args.seed + gi, ckpt_dirs=sw.prior_checkpoint_dirs(cfg, gender))
        st = np.stack([np.stack([pairs[s][k]["wpath"] for s in idx]) for k in range(2)])          # [2,S,2,3]
        agent_seeds = [[pairs[s][k]["seed"] for s in idx] for k in range(2)]
        grp = CrowdGroupEnv(S, st, body, prior, vposer, cfg=sw.env_cfg_from_yaml(cfg), seed=args.seed + 100 * local_rank + gi,
                            keep_rollout=True, scene_rings=sampler.rings, static_scene=static_scene, agent_seeds=agent_seeds,
                            vp_thresh=14.0, goal_terminates=False, gender=gender)
        obs = grp.reset()
        episodes = [[[] for _ in range(S)] for _ in range(2)]
        ep_ret = torch.zeros(2, S, device="cuda")
        pol_out = [dict(), dict()]
        wpath0 = [m.wpath.cpu() for m in grp.members]
        max_depth = grp.members[0].cfg["max_depth"]
        for _ in range(max_depth):                      # only max_depth terminates (crowd_env_egobody_eval.py:378)
            for k, m in enumerate(grp.members):
                out = policy(obs[k], out=pol_out[k])
                o, rew, term = m.step(out["act"], auto_reset=False)
                ep_ret[k] += rew
                mb, pp, fr = m.marker_b.cpu(), m.pred_params.cpu(), m.prev_frame.cpu()
                pel = m.joints.reshape(S, 20, -1, 3)[:, :, 0].cpu()
                for s in range(S):
                    episodes[k][s].append([mb[s:s + 1], pp[s:s + 1], m.betas[s].cpu(), gender, fr[s, :9].reshape(3, 3),
                                           fr[s, 9:].reshape(1, 3), pel[s:s + 1], "2-frame"])
                obs[k] = o
        bad = grp.invalid().cpu().numpy()
        for s in range(S):
            gs = idx[s]
            if bad[s]:
                why = " + ".join(n for b, n in ((1, "invalid pelvis location"), (2, "unrealistic pose")) if bad[s] & b)
                print(f"scene {gs}: {'kept although flagged' if keep_invalid else 'dropped'} ({why})")
                results["dropped"].append(gs)
                if not keep_invalid:
                    continue
            for k in range(2):
                man_id = str(k) if num_scenes == 1 else f"s{gs}_{k}"
                scene = {"wpath": wpath0[k][s], "navmesh_path": pairs[gs][k]["navmesh_path"], "scene_path": pairs[gs][k]["scene_path"]}
                save_rollout_results(scene, episodes[k][s], out_dir, man_id=man_id)
                results["rew"].append(float(ep_ret[k, s]))
                results["len"].append(max_depth)
            results["written"].append(gs)
    if results["rew"]:
        print(f'Final reward: {np.mean(results["rew"])}, length: {np.mean(results["len"])}')
    print(f'{len(results["written"])} scene(s) written to {out_dir}, {len(results["dropped"])} dropped')
    return results


if __name__ == "__main__":
    extra = [("--scene-name", dict(type=str, default="cab_e")),
             ("--dynamic-holes", dict(action="store_true",
                                      help="make the other person's marker box a hole of the walkable polygon (the reference's "
                                           "Polygon(scene_poly, holes) call does not, see DESIGN.md)")),
             ("--keep-invalid", dict(action="store_true", help="write sequences that trip a data filter too (untrained weights)"))]
    a = get_args(extra=extra)
    main(a, num_scenes=a.num_scenes or 1, scene_name=a.scene_name, static_scene=not a.dynamic_holes, keep_invalid=a.keep_invalid)
