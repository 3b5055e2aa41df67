"""Wiring shared by crowd_ppo/main_ppo.py, main_ppo_box.py and bench.py: assets -> operators -> envs -> policy.

Follows the `__main__` blocks of the reference drivers (crowd_ppo/main_ppo.py:246-309, main_ppo_box.py:255-312) and
`load_model` (crowd_ppo/primitive_model.py:74-96).  Licensed assets (SMPL-X npz, VPoser snapshot, C-VAE / regressor
checkpoints, room0_sdf.pkl, box scene set) are used when present under the reference's paths (cwd = motion/);
otherwise the seeded synthetic stand-ins of `egogen_amd.synth` are built (BASELINE.json: "synthetic random-init
bodies/scenes").
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

from . import synth
from .body_model import BodyModelHandle
from .crowd_env import VecCrowdEnv
from .models import (ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, GAMMAPrimitiveCombo, POLICY_CFG, PREDICTOR_CFG,
                     REGRESSOR_CFG, VPoserEncoder)
from .ppo_policy import GAMMAPPOPolicy

RESULTS_ROOT = os.path.join("results", "crowd_ppo")


def create_dirs(cfg_name: str, run_name: str = "collision_test") -> Dict[str, str]:
    """primitive_model.py:41-54: results/crowd_ppo/<cfg_name>/<wandb.name>/{results,checkpoints,logs}"""
    exp = os.path.join(RESULTS_ROOT, cfg_name, run_name)
    d = {"cfg_exp_dir": exp, "cfg_result_dir": os.path.join(exp, "results"), "cfg_ckpt_dir": os.path.join(exp, "checkpoints"),
         "cfg_log_dir": os.path.join(exp, "logs")}
    for p in list(d.values())[1:]:
        os.makedirs(p, exist_ok=True)
    return d


_PKG_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cfg_samp20")
CFG_DIR = os.path.join("crowd_ppo", "cfg_samp20")   # the reference's location, relative to the working directory (motion/)


def _read_yaml(name: str, cfg_dir: Optional[str] = None) -> dict:
    import yaml
    for d in ([cfg_dir] if cfg_dir else [CFG_DIR, _PKG_CFG_DIR]):
        f = os.path.join(d, name)
        if os.path.exists(f):
            with open(f, "r") as fh:
                cfg = yaml.safe_load(fh)
            if not isinstance(cfg, dict):
                raise ValueError(f"{f}: expected a mapping at the top level")
            return cfg
    raise FileNotFoundError(f"config {name!r} not found under {cfg_dir or CFG_DIR + ' or ' + _PKG_CFG_DIR}")


def load_model(box: bool = False, cfg_dir: Optional[str] = None) -> dict:
    """crowd_ppo/primitive_model.py:74-96 `load_model`, configuration half: read
    cfg_samp20/MPVAEPolicy_samp_collision(_2).yaml (plain yaml instead of OmegaConf), create
    results/crowd_ppo/<cfg_name>/<wandb.name>/{results,checkpoints,logs}, set trainconfig.save_dir / log_dir and write the
    resolved configuration to <exp_dir>/config.yaml (:78-82).  The motion-prior half (configure_model, :56-72) is
    `build_motion_prior`, whose checkpoint directories come from the combo config named by trainconfig.cfg_2frame_male.
    Raises FileNotFoundError where the reference calls sys.exit() (:19-21)."""
    import yaml
    cfg = _read_yaml("MPVAEPolicy_samp_collision_2.yaml" if box else "MPVAEPolicy_samp_collision.yaml", cfg_dir)
    for sec in ("modelconfig", "lossconfig", "trainconfig"):
        if not isinstance(cfg.get(sec), dict):
            raise KeyError(f"config has no {sec!r} section")
    run = (cfg.get("wandb") or {}).get("name", "collision_test")
    dirs = create_dirs(cfg["cfg_name"], run)
    cfg.update(dirs)
    cfg["trainconfig"]["save_dir"] = dirs["cfg_ckpt_dir"]
    cfg["trainconfig"]["log_dir"] = dirs["cfg_log_dir"]
    if int(os.environ.get("RANK", "0")) == 0:   # data-parallel ranks share the directory: one writer
        with open(os.path.join(dirs["cfg_exp_dir"], "config.yaml"), "w") as fh:
            yaml.safe_dump(cfg, fh, sort_keys=False)
    return cfg


def env_cfg_from_yaml(cfg: dict) -> dict:
    """The fields CrowdEnv.step / reset read from the yaml (crowd_env_2f.py:151-152,235,268-281,331; crowd_env_2f_box.py:285-303)
    as the flat dict VecCrowdEnv takes."""
    from .crowd_env import DEFAULT_CFG
    m, l, t = cfg["modelconfig"], cfg["lossconfig"], cfg["trainconfig"]
    out = dict(DEFAULT_CFG)
    out.update(reproj_factor=float(m["reproj_factor"]), map_res=int(m.get("map_res", 16)), map_extent=float(m.get("map_extent", 0.8)),
               goal_thresh=float(t["goal_thresh"]), max_depth=int(t["max_depth"]), pene_thres=float(t.get("pene_thres", 3)),
               pene_type=str(l.get("pene_type", "body")))
    for k in ("weight_vp", "weight_floor", "weight_skate", "weight_target_dist", "weight_face_target", "weight_look_target",
              "weight_pene", "weight_success"):
        out[k] = float(l[k])
    return out


def policy_cfg_from_yaml(cfg: dict) -> dict:
    """modelconfig of the policy yaml -> constructor config of GAMMAPolicyBase / GAMMAActor / GAMMACritic (main_ppo.py:108-113)."""
    out = dict(POLICY_CFG)
    out.update({k: cfg["modelconfig"][k] for k in POLICY_CFG if k in cfg["modelconfig"]})
    return out


def prior_checkpoint_dirs(cfg: dict, gender: str = "male", cfg_dir: Optional[str] = None, ckpt_root: str = RESULTS_ROOT):
    """configure_model (primitive_model.py:56-72): trainconfig.cfg_2frame_<gender> names the combo config whose modelconfig
    names the predictor / regressor configs; their checkpoints live under results/crowd_ppo/<name>/checkpoints."""
    combo = _read_yaml(cfg["trainconfig"][f"cfg_2frame_{gender}"] + ".yml", cfg_dir)
    mc = combo["modelconfig"]
    return (os.path.join(ckpt_root, mc["predictor_config"], "checkpoints"), os.path.join(ckpt_root, mc["regressor_config"], "checkpoints"))


def load_body_model(gender: str = "male", seed: int = 0, num_verts: int = synth.NUM_VERTS, model_dir: str = "data/smplx/models"):
    """Real SMPLX_<GENDER>.npz when available (smplx.create(...) arguments of baseops.py:291-320), else synthetic."""
    path = os.path.join(model_dir, "smplx", f"SMPLX_{gender.upper()}.npz")
    if os.path.exists(path) and num_verts == synth.NUM_VERTS:
        return _load_real_smplx(path), True
    return synth.make_body_model(seed, num_verts=num_verts), False


def _load_real_smplx(path: str) -> Dict[str, np.ndarray]:
    """Field preparation of smplx.SMPLX.__init__ [upstream smplx 0.1.28]: 10 betas, posedirs reshaped to
    [486, 3V], first 12 hand PCA components, hand means (flat_hand_mean=False), static face landmarks,
    vertex-selected extra joints (smplx/vertex_ids.py 'smplx' table)."""
    d = np.load(path, allow_pickle=True, encoding="latin1")
    V = d["v_template"].shape[0]
    posedirs = np.asarray(d["posedirs"]).reshape(V * 3, -1).T
    parents = np.asarray(d["kintree_table"][0]).astype(np.int32).copy()
    parents[0] = -1
    faces = np.asarray(d["f"]).astype(np.int64)
    vid = {"nose": 9120, "reye": 9929, "leye": 9448, "rear": 616, "lear": 6, "rthumb": 8079, "rindex": 7669, "rmiddle": 7794,
           "rring": 7905, "rpinky": 8022, "lthumb": 5361, "lindex": 4933, "lmiddle": 5058, "lring": 5169, "lpinky": 5286,
           "LBigToe": 5770, "LSmallToe": 5780, "LHeel": 8846, "RBigToe": 8463, "RSmallToe": 8474, "RHeel": 8635}
    order = ["nose", "reye", "leye", "rear", "lear", "LBigToe", "LSmallToe", "LHeel", "RBigToe", "RSmallToe", "RHeel"]
    for hand in "lr":
        order += [hand + t for t in ("thumb", "index", "middle", "ring", "pinky")]
    lmk_faces = np.asarray(d["lmk_faces_idx"]).astype(np.int64)
    return {
        "v_template": np.asarray(d["v_template"], np.float32),
        "shapedirs": np.asarray(d["shapedirs"][:, :, :10], np.float32),
        "posedirs": np.ascontiguousarray(posedirs, np.float32),
        "J_regressor": np.asarray(d["J_regressor"], np.float32),
        "parents": parents,
        "lbs_weights": np.asarray(d["weights"], np.float32),
        "hand_comps_l": np.asarray(d["hands_componentsl"][:12], np.float32),
        "hand_comps_r": np.asarray(d["hands_componentsr"][:12], np.float32),
        "hand_mean_l": np.asarray(d["hands_meanl"], np.float32),
        "hand_mean_r": np.asarray(d["hands_meanr"], np.float32),
        "extra_vids": np.array([vid[k] for k in order], np.int32),
        "lmk_vids": faces[lmk_faces].astype(np.int32),
        "lmk_bary": np.asarray(d["lmk_bary_coords"], np.float32),
    }


def build_motion_prior(device="cuda", seed: int = 0, ckpt_root: str = RESULTS_ROOT, ckpt_dirs=None) -> GAMMAPrimitiveCombo:
    """GAMMAPrimitiveComboGenOP.build_model (models_GAMMA_primitive.py:1116-1148): predictor epoch-400.ckp (else
    epoch-200.ckp), regressor epoch-100.ckp, key 'model_state_dict'.  Random init (seeded) when the files are absent."""
    torch.manual_seed(seed)
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    pdir = os.path.join(ckpt_root, "MPVAE_samp20_2frame_rollout", "checkpoints")
    rdir = os.path.join(ckpt_root, "MoshRegressor_v3_male", "checkpoints")
    if ckpt_dirs is not None:
        pdir, rdir = ckpt_dirs
    for name in ("epoch-400.ckp", "epoch-200.ckp"):
        f = os.path.join(pdir, name)
        if os.path.exists(f):
            combo.predictor.load_state_dict(torch.load(f, map_location="cpu")["model_state_dict"])
            break
    f = os.path.join(rdir, "epoch-100.ckp")
    if os.path.exists(f):
        combo.regressor.load_state_dict(torch.load(f, map_location="cpu")["model_state_dict"])
    return combo.to(device).eval()


def build_vposer(device="cuda", seed: int = 0, model_dir: str = "data/smplx/models") -> VPoserEncoder:
    torch.manual_seed(seed + 17)
    enc = VPoserEncoder()
    snap_dir = os.path.join(model_dir, "vposer_v1_0", "snapshots")
    if os.path.isdir(snap_dir):
        snaps = sorted(f for f in os.listdir(snap_dir) if f.endswith(".pt"))
        if snaps:
            enc.load_state_dict(torch.load(os.path.join(snap_dir, snaps[-1]), map_location="cpu"), strict=False)
    else:
        with torch.no_grad():  # non-trivial BatchNorm statistics for the synthetic encoder
            enc.bodyprior_enc_bn1.running_mean.normal_(0, 0.05)
            enc.bodyprior_enc_bn1.running_var.uniform_(0.5, 1.5)
            enc.bodyprior_enc_bn2.running_mean.normal_(0, 0.05)
            enc.bodyprior_enc_bn2.running_var.uniform_(0.5, 1.5)
    return enc.to(device).eval()


def build_policy(args, device="cuda", policy_cfg: Optional[dict] = None) -> GAMMAPPOPolicy:
    """crowd_ppo/main_ppo.py:108-162: nets, orthogonal(sqrt 2) init of every nn.Linear, last-policy-layer x0.01,
    AdamW(lr, weight_decay 0.01), GAMMAPPOPolicy."""
    torch.manual_seed(args.seed)
    pc = policy_cfg or POLICY_CFG
    actor, critic, shared_net = GAMMAActor(pc), GAMMACritic(pc), GAMMAPolicyBase(pc)
    actor_critic = ActorCritic(actor, critic, shared_net)
    for m in actor_critic.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            torch.nn.init.zeros_(m.bias)
    for m in actor_critic.actor.pnet.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.zeros_(m.bias)
            m.weight.data.copy_(0.01 * m.weight.data)
    actor_critic.to(device)
    graph = bool(getattr(args, "update_graph", False))
    if graph:
        try:  # one fused multi-tensor kernel per step instead of ~10 foreach kernels
            optim = torch.optim.AdamW(actor_critic.parameters(), lr=args.lr, weight_decay=0.01, capturable=True, fused=True)
        except Exception:
            optim = torch.optim.AdamW(actor_critic.parameters(), lr=args.lr, weight_decay=0.01, capturable=True, foreach=True)
    else:
        optim = torch.optim.AdamW(actor_critic.parameters(), lr=args.lr, weight_decay=0.01)
    policy = GAMMAPPOPolicy(actor, critic, shared_net, optim, None, discount_factor=args.gamma, gae_lambda=args.gae_lambda,
                            max_grad_norm=args.max_grad_norm, vf_coef=args.vf_coef, ent_coef=args.ent_coef,
                            weight_kld=args.weight_kld, reward_normalization=args.rew_norm, eps_clip=args.eps_clip,
                            value_clip=args.value_clip, dual_clip=args.dual_clip, advantage_normalization=args.norm_adv,
                            recompute_advantage=args.recompute_adv, deterministic_eval=args.deterministic_eval, seed=args.seed,
                            use_update_graph=graph,
                            **({"update_precision": args.update_precision} if getattr(args, "update_precision", None) else {}))
    return policy


def load_scene_file(path: str) -> dict:
    """A scene prepared with `egogen_amd.scene_gen` (`save_scene`): with an SDF grid -> the room kind (SDF penetration term,
    walkable polygon for the egosensing rays, start / target pairs); without -> one scene of the box kind (walkability map)."""
    with np.load(path) as z:
        d = {k: z[k] for k in z.files}
    off = d["ring_off"]
    rings = [d["ring_xy"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    if "sdf_sdf" in d:
        return dict(scene_kind="sdf", sdf_dict={"sdf": d["sdf_sdf"], "center": d["sdf_center"], "scale": d["sdf_scale"]}, rings=rings,
                    pairs=d["pairs"])
    return dict(scene_kind="box", box_scenes=[{"edges": d["edges"], "tris": d["tris"], "floor_height": float(d["floor_height"]),
                                               "pairs": d["pairs"]}])


def build_scene(kind: str, sdf_res: int = 256, seed: int = 0, data_dir: str = "data"):
    """kind: 'room0' (Replica room0 polygon + pairs, SDF from data/room0_sdf.pkl if present else a synthetic room0-shaped
    grid), 'single_box' (BASELINE config 2), 'box' (random_box_obstacle_new stand-in), or the path of a `.npz` scene
    prepared with `egogen_amd.scene_gen.save_scene` (new scenes)."""
    if kind.endswith(".npz"):
        return load_scene_file(kind)
    if kind == "box":
        return dict(scene_kind="box", box_scenes=synth.make_box_scenes(64, 2048, seed=seed))
    if kind == "room0":
        sdf_path = os.path.join(data_dir, "room0_sdf.pkl")
        if os.path.exists(sdf_path):
            sdf = np.load(sdf_path, allow_pickle=True)
            sdf = {k: np.asarray(v, np.float32) for k, v in sdf.items()}
        else:
            sdf = synth.make_sdf_scene(sdf_res, room="room0", seed=seed)
        A = synth.load_assets()
        return dict(scene_kind="sdf", sdf_dict=sdf, rings=synth.room0_polygon(), pairs=A["room0_pairs"])
    if kind == "single_box":
        sdf = synth.make_sdf_scene(sdf_res, room="single_box", seed=seed)
        rng = np.random.default_rng(seed + 3)
        pairs = np.zeros((4096, 2, 3), np.float32)
        n = 0
        while n < len(pairs):
            c = rng.uniform(-3.3, 3.3, (8192, 2, 2))
            ok = np.linalg.norm(c[:, 0] - c[:, 1], axis=1) >= 1.7
            good = c[ok][: len(pairs) - n]
            pairs[n:n + len(good), :, :2] = good
            n += len(good)
        return dict(scene_kind="sdf", sdf_dict=sdf, rings=synth.sdf_scene_polygon(sdf), pairs=pairs)
    raise ValueError(f"unknown scene {kind!r}")


def build_env(num_agents: int, scene: dict, body: BodyModelHandle, prior: GAMMAPrimitiveCombo, vposer: VPoserEncoder,
              finetuning=False, seed=0, keep_rollout=False, use_graph=False, cfg: Optional[dict] = None) -> VecCrowdEnv:
    return VecCrowdEnv(num_agents, body, prior, vposer, finetuning=finetuning, seed=seed, keep_rollout=keep_rollout,
                       use_graph=use_graph, cfg=cfg, **scene)
