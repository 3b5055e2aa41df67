"""Shared test helpers (no GPU needed to import)."""
import ast
import os

import numpy as np
import torch

from egogen_amd.synth import seeded_fill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rebuild_state_dict(g, seeds, prefixes, gains=None, dtype=torch.float32):
    """Rebuild the seeded weights the reference module was filled with (scripts/gen_goldens.py)."""
    keys = [str(k) for k in g["state_dict_keys"]]
    shapes = [ast.literal_eval(str(s)) for s in g["state_dict_shapes"]]
    out = {}
    gains = gains or [1.0] * len(seeds)
    if len(prefixes) == 1 and prefixes[0] == "":
        groups = [("", list(zip(keys, shapes)))]
    else:
        groups = [(p, [(k[len(p):], s) for k, s in zip(keys, shapes) if k.startswith(p)]) for p in prefixes]
    for (p, items), seed, gain in zip(groups, seeds, gains):
        vals = seeded_fill(dict(items), int(seed), gain=gain)
        for k, v in vals.items():
            out[p + k] = torch.from_numpy(v).to(dtype)
    return out


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# ---------------------------------------------------------------------------------------------
# shared synthetic "world" for env / trainer tests
# ---------------------------------------------------------------------------------------------

def seeded_prior_state_dict(seed=200, pred_gain=0.7, reg_gain=0.6):
    """Seeded weights for GAMMAPrimitiveCombo (keys of the reference state_dict)."""
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    sd = combo.state_dict()
    pk = {k: tuple(v.shape) for k, v in sd.items() if k.startswith("predictor.")}
    rk = {k: tuple(v.shape) for k, v in sd.items() if k.startswith("regressor.")}
    out = {k: torch.from_numpy(v) for k, v in seeded_fill(pk, seed, gain=pred_gain).items()}
    out.update({k: torch.from_numpy(v) for k, v in seeded_fill(rk, seed + 1, gain=reg_gain).items()})
    return out


def seeded_vposer_state_dict(seed=210):
    from egogen_amd.models import VPoserEncoder
    enc = VPoserEncoder()
    vals = seeded_fill({k: tuple(v.shape) for k, v in enc.state_dict().items()}, seed)
    out = {k: torch.from_numpy(v) for k, v in vals.items()}
    for k in list(out):
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(0)
    return out


def build_world(V=1536, A=6, scene_kind="sdf", sdf_res=48, n_pairs=64, n_scenes=4, finetuning=False, gpu=True, seed=0):
    """Body model + scene + nets, as a GPU VecCrowdEnv (if gpu) and the CPU OracleCrowdEnv on the same assets."""
    from egogen_amd import synth
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    bm = synth.make_body_model(seed, num_verts=V)
    mk, feet, fmi = synth.marker_ids(V), synth.feet_vids(V), synth.feet_marker_idx()
    prior_sd, vposer_sd = seeded_prior_state_dict(), seeded_vposer_state_dict()
    w = {"bm": bm, "mk": mk, "feet": feet, "prior_sd": prior_sd, "vposer_sd": vposer_sd, "A": A}
    rng = np.random.default_rng(seed + 5)
    if scene_kind == "sdf":
        scene = synth.make_sdf_scene(sdf_res)
        rings = synth.sdf_scene_polygon(scene)
        pairs = np.zeros((n_pairs, 2, 3), np.float32)
        pairs[:, :, :2] = rng.uniform(-3.0, 3.0, (n_pairs, 2, 2))
        w.update(scene=scene, rings=rings, pairs=pairs)
        sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
        okw = dict(scene_kind="sdf", sdf_dict=sd, edges=synth.rings_to_edges(rings))
        gkw = dict(scene_kind="sdf", sdf_dict=scene, rings=rings, pairs=pairs)
    else:
        scenes = synth.make_box_scenes(n_scenes, n_pairs, seed=seed + 7)
        w.update(box_scenes=scenes)
        okw = dict(scene_kind="box", box_scenes=scenes)
        gkw = dict(scene_kind="box", box_scenes=scenes)
    w["oracle"] = OracleCrowdEnv(BodyModel(bm), prior_sd, {k: v.float() for k, v in vposer_sd.items()}, mk, feet, fmi,
                                 finetuning=finetuning, **okw)
    if gpu:
        from egogen_amd.body_model import BodyModelHandle
        from egogen_amd.crowd_env import VecCrowdEnv
        from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder
        h = BodyModelHandle(bm, mk, feet)
        combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
        combo.load_state_dict(prior_sd)
        combo.cuda().eval()
        vp = VPoserEncoder()
        vp.load_state_dict(vposer_sd)
        vp.cuda().eval()
        w["env"] = VecCrowdEnv(A, h, combo, vp, finetuning=finetuning, seed=seed, **gkw)
        w["handle"] = h
        w["oracle"].level_set_band = level_set_band()   # band of `pene_near_zero` (the HIP path's counts are compared inside it)
    return w


def level_set_band():
    """Distance from the SDF's zero level set (metres) inside which a penetration count of the HIP path may differ from a CPU
    evaluation: 2e-5 (fp32 round-off of the vertex chain) in EVERY blend mode - mode 3 ("f16mix", the library default) classifies
    with its fp16 product and re-evaluates in fp32 what that product cannot decide (csrc/body_model.hip: lbs_fix_process)."""
    return 2e-5


def free_port() -> int:
    """A TCP port the kernel just handed out on 127.0.0.1 (rendezvous of the spawned multi-rank tests: a port derived from the pid
    can collide with a listener of an earlier test still in TIME_WAIT, or with another run on the same box)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
