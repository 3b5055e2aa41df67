#!/usr/bin/env python3
"""Yardstick for the absolute floors of the env parity tests (tests/test_env_gpu.py::_close): what fp32 itself costs.

The CPU oracle is run twice from the same starts and the same actions - in float32 (the arithmetic of the reference's
PyTorch path) and in float64 - over a reset and one step, for the SDF scene and the box scene set at the sizes the tests use.
Per compared quantity the largest |fp32 - fp64| entry is the yardstick: an fp32 evaluation, in ANY operation order, sits
about that far from the true value, so a GPU-vs-oracle bound below it would test the operation order, not the result.  The
tests bound every entry by 1e-4 |b| + floor with floor = 3 x yardstick of the quantity's kind (rounded up).
Writes profiles/r04_env_tolerances.txt.  CPU only (no HIP device needed)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from egogen_amd import synth                                    # noqa: E402
from oracle.env import OracleCrowdEnv                            # noqa: E402
from oracle.smplx_lbs import BodyModel                           # noqa: E402
from tests.helpers import build_world                            # noqa: E402

KIND = {"state": "m", "seed transl": "m", "seed pose": "rad", "R0": "unit", "T0": "m", "dist": "m", "wpath": "m", "Y_gen": "m",
        "pred transl": "m", "joints": "m", "markers_proj": "m", "egosensing": "unit", "obs dist": "unit", "reward": "reward",
        "r_skate": "reward", "r_floor": "reward", "r_face": "reward", "r_look": "reward", "r_target_dist": "reward", "r_vp": "reward"}


def run(scene_kind, dtype, A, seed):
    w = build_world(A=A, scene_kind=scene_kind, gpu=False, n_pairs=32, n_scenes=3)
    o = w["oracle"]
    if dtype == torch.float64:
        if scene_kind == "sdf":
            sd = {k: torch.as_tensor(np.asarray(w["scene"][k])).double() for k in ("sdf", "center", "scale")}
            kw = dict(scene_kind="sdf", sdf_dict=sd, edges=synth.rings_to_edges(w["rings"]))
        else:
            kw = dict(scene_kind="box", box_scenes=w["box_scenes"])
        o = OracleCrowdEnv(BodyModel(w["bm"], dtype=torch.float64), w["prior_sd"], {k: v.float() for k, v in w["vposer_sd"].items()},
                           w["mk"], w["feet"], synth.feet_marker_idx(), **kw)
    rng = np.random.default_rng(seed)
    ms = synth.load_assets()
    nv = len(ms["seed_poses"]) - 1
    if scene_kind == "sdf":
        pairs = w["pairs"][rng.integers(0, len(w["pairs"]), A)]
        variant, yaw, scene = [5] * A, None, None
    else:
        scene = rng.integers(0, 3, A)
        pairs = np.stack([w["box_scenes"][s]["pairs"][rng.integers(0, 32)] for s in scene])
        variant, yaw = rng.integers(0, nv, A), (rng.uniform(-1, 1, A) * 2 * np.pi * 0.1).astype(np.float32)
    poses = torch.tensor(np.stack([ms["seed_poses"][s:s + 2, :66] for s in variant]), dtype=torch.float32)
    trans = torch.tensor(np.stack([ms["seed_trans"][s:s + 2] for s in variant]), dtype=torch.float32)
    betas = torch.tensor(ms["seed_betas"], dtype=torch.float32).reshape(1, 10).repeat(A, 1)
    tr, go, bp, wp = o.next_body(torch.as_tensor(pairs[:, 0]), torch.as_tensor(pairs[:, 1]), poses, trans, betas,
                                 yaw_jitter=None if yaw is None else torch.as_tensor(yaw))
    obs0, _ = o.reset_from(tr, go, bp, betas, wp, scene_idx=scene)
    out = {"state": o.state.clone(), "seed transl": o.body_param_seed[..., :3].clone(), "seed pose": o.body_param_seed[..., 6:].clone(),
           "R0": o.R0.clone(), "T0": o.T0.clone(), "dist": o.dist.clone(), "wpath": o.wpath.clone(), "egosensing": obs0["egosensing"].clone()}
    z = torch.randn(A, 128, generator=torch.Generator().manual_seed(seed)).to(o.dt)
    obs, rew, _ = o.step(z)
    L = o.last
    out.update({"Y_gen": L["Y_gen"], "pred transl": L["pred_params"][..., :3], "joints": L["joints"], "markers_proj": L["markers_proj"],
                "reward": rew, "obs dist": obs["dist"], "egosensing (step)": obs["egosensing"]})
    for k in ("r_skate", "r_floor", "r_face", "r_look", "r_target_dist", "r_vp"):
        out[k] = L[k]
    return {k: torch.as_tensor(v).double() for k, v in out.items()}


def main():
    worst = {}
    rows = []
    for scene_kind in ("sdf", "box"):
        for seed in (0, 1, 2):
            a32, a64 = run(scene_kind, torch.float32, 6, seed), run(scene_kind, torch.float64, 6, seed)
            for k in a32:
                d = float((a32[k] - a64[k]).abs().max())
                rows.append((scene_kind, seed, k, float(a64[k].abs().max()), d))
                kk = KIND.get(k.replace(" (step)", ""), "m")
                worst[kk] = max(worst.get(kk, 0.0), d)
    lines = ["# fp32 oracle vs float64 oracle, same starts and actions, reset + one step, A = 6, V = 1536 (scripts/env_tolerance_yardstick.py)",
             "# scene  seed  quantity              max|fp64|   max|fp32 - fp64|"]
    lines += [f"  {s:<5s} {sd:4d}  {k:<20s} {m:10.3e}  {d:10.3e}" for s, sd, k, m, d in rows]
    lines.append("# largest |fp32 - fp64| per kind of quantity -> floor of tests/test_env_gpu.py::_close = 3 x this, rounded up:")
    for kk, d in sorted(worst.items()):
        lines.append(f"  {kk:<8s} {d:10.3e}")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r04_env_tolerances.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[-6:]))


if __name__ == "__main__":
    main()
