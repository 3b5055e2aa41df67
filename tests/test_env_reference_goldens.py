"""The env composite against the REFERENCE'S OWN CODE, executed: tests/golden/env_step_ref.npz and env_box_ref.npz hold what
crowd_env_2f.CrowdEnv / crowd_env_2f_box.CrowdEnv (reset: sampler next_body -> _canonicalize_2frame -> start check -> features;
step: motion prior -> blend -> SMPL-X -> reward block -> re-canonicalisation -> termination) returned when
scripts/gen_env_goldens.py ran them in the build container (their smplx / rotation / VPoser / shapely calls served by the
oracle's restatements, everything else the reference's lines).

  * CPU (`-m "not gpu"`): oracle/env.py::OracleCrowdEnv - until round 5 a restatement pinned by reading - reproduces every
    recorded quantity: sampler output, reset state and observation, and per step the 18 predicted frames, blended parameters,
    127 joints, blended markers, the eight reward terms, the penetration counts (exact), termination, the new state / seed /
    frame, the observation.
  * GPU (`-m gpu`): the HIP path (VecCrowdEnv through the C ABI) against the same fixtures.

Cases (sdf env): a free walk of three steps; an obstacle on the agent with and without finetuning (>= 40 vertices: terminates
only when finetuning); a grazing obstacle (0 < count < 40: no termination, 0 < r_pene < 1); goal reached; depth limit; an
implausible pose (VPoser norm > 11).  Box env: free steps, a start on the obstacle (reset rejects it), a step whose marker
box covers non-walkable cells.
"""
import json

import numpy as np
import pytest
import torch

from tests.helpers import load_golden, seeded_prior_state_dict, seeded_vposer_state_dict

TOL = 1e-4
# absolute floors next to north_star's 1e-4 relative: what two fp32 evaluations of the same chain differ by (metre-scale
# coordinates ~ 1e-6 per operation, a few hundred operations deep; unit vectors / rotations; normalised egosensing)
FLOOR = {"m": 2e-5, "unit": 2e-5, "ego": 3.6e-4, "reward": 2e-5}


def _close(a, b, kind, what):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    d = np.abs(a - b)
    bound = TOL * np.abs(b) + FLOOR[kind]
    bad = d > bound
    assert not bad.any(), f"{what}: {int(bad.sum())} of {d.size} outside 1e-4 |ref| + {FLOOR[kind]:g}; worst |d| {d.max():.3e}"


def _aa_close(a, b, what):
    from oracle.rot import tgm_angle_axis_to_rotation_matrix as aa2R
    f = lambda x: torch.as_tensor(np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, np.float32)).reshape(-1, 3)
    _close(aa2R(f(a)), aa2R(f(b)), "unit", what)


def _vposer_sd(gain):
    sd = {k: v.float() for k, v in seeded_vposer_state_dict().items()}
    if gain != 1.0:
        for k in list(sd):
            if k.startswith("bodyprior_enc_mu."):
                sd[k] = sd[k] * gain
    return sd


def _motion_seed(start=5):
    from egogen_amd import synth
    ms = synth.load_assets()
    poses = torch.tensor(np.asarray(ms["seed_poses"])[start:start + 2, :66], dtype=torch.float32)[None]
    trans = torch.tensor(np.asarray(ms["seed_trans"])[start:start + 2], dtype=torch.float32)[None]
    betas = torch.tensor(np.asarray(ms["seed_betas"]), dtype=torch.float32).reshape(1, 10)
    return poses, trans, betas


@pytest.fixture(scope="module")
def world():
    from egogen_amd import synth
    g = load_golden("env_step_ref.npz")
    assert int(g["body_model_seed"]) == 0
    bm = synth.make_body_model(0)
    return {"g": g, "bm": bm, "mk": synth.marker_ids(), "feet": synth.feet_vids(), "fmi": synth.feet_marker_idx(),
            "prior_sd": seeded_prior_state_dict(int(g["prior_seed"]), *[float(v) for v in g["prior_gains"]]),
            "res": int(g["sdf_res"]), "cfg": json.loads(str(g["cfg_json"]))}


def _sdf_tensors(scene):
    return {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}


def _check_step_common(g, pre, got, counts=None, w_pene=1.0, band_key="pene_near_zero"):
    """`got`: dict of the quantities of one step, named like the fixture's."""
    _close(got["Y_gen"], g[pre + "Y_gen"], "m", pre + "Y_gen")
    _close(got["pred_params"][:, :3], g[pre + "pred_params"][:, :3], "m", pre + "pred transl")
    _aa_close(got["pred_params"][:, 3:6], g[pre + "pred_params"][:, 3:6], pre + "pred glorot")
    _close(got["pred_params"][:, 6:], g[pre + "pred_params"][:, 6:], "unit", pre + "pred pose")
    _close(got["joints"], g[pre + "joints"], "m", pre + "joints")
    if "marker_b" in got:
        _close(got["marker_b"], g[pre + "marker_b"], "m", pre + "marker_b")
    for k in ("r_skate", "r_floor", "r_face_target", "r_look_target", "r_goal", "r_target_dist", "r_vp"):
        _close(np.float64(got[k]), g[pre + k], "reward", pre + k)
    slack = 0.0
    if counts is not None:
        # integer counts: exact, except for vertices the REFERENCE's own evaluation put within fp32 round-off (2e-5 m) of the zero
        # level set; r_pene = exp(-sum(count) / 20 / 10) and the reward inherit exactly that slack, nothing is added
        near = g[pre + band_key]
        dcnt = np.abs(np.asarray(counts, np.int64) - g[pre + "pene_count"])
        assert (dcnt <= near).all(), (counts, g[pre + "pene_count"], near)
        slack = float(near.sum()) / 200.0
    d = abs(float(got["r_pene"]) - float(g[pre + "r_pene"]))
    assert d <= slack * float(g[pre + "r_pene"]) * 1.01 + TOL * float(g[pre + "r_pene"]) + FLOOR["reward"], (pre + "r_pene", d, slack)
    d = abs(float(got["reward"]) - float(g[pre + "reward"]))
    assert d <= w_pene * slack * float(g[pre + "r_pene"]) * 1.01 + TOL * abs(float(g[pre + "reward"])) + FLOOR["reward"], (pre + "reward", d)
    assert bool(got["terminated"]) == bool(g[pre + "terminated"]), pre + "terminated"
    assert not bool(g[pre + "truncated"])
    _close(got["after_state"], g[pre + "after_state"], "unit", pre + "state")
    _close(got["after_seed"][:, :3], g[pre + "after_seed"][:, :3], "m", pre + "seed transl")
    _aa_close(got["after_seed"][:, 3:6], g[pre + "after_seed"][:, 3:6], pre + "seed glorot")
    _close(got["after_seed"][:, 6:], g[pre + "after_seed"][:, 6:], "unit", pre + "seed pose")
    _close(got["after_R0"], g[pre + "after_R0"], "unit", pre + "R0")
    _close(np.asarray(got["after_T0"]).reshape(-1), g[pre + "after_T0"].reshape(-1), "m", pre + "T0")
    _close(np.asarray(got["after_dist"]).reshape(-1), g[pre + "after_dist"].reshape(-1), "m", pre + "dist")
    _close(got["obs_ego"], g[pre + "obs_ego"], "ego", pre + "egosensing")
    _close(np.asarray(got["obs_dist"]).reshape(-1), g[pre + "obs_dist"].reshape(-1), "unit", pre + "obs dist")
    _close(np.asarray(got["obs_time"]).reshape(-1), g[pre + "obs_time"].reshape(-1), "unit", pre + "obs time")
    _close(got["after_state"], g[pre + "obs_state"], "unit", pre + "obs state")


SDF_CASES = ["free", "pene_ft", "pene", "graze_ft", "goal", "depth", "vposer"]


@pytest.mark.parametrize("case", SDF_CASES)
def test_oracle_env_matches_reference_execution(world, case):
    from egogen_amd import synth
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    g = world["g"]
    assert case in [str(c) for c in g["cases"]]
    pre = case + "_"
    scene = synth.make_sdf_scene(world["res"])
    edges = synth.rings_to_edges(synth.sdf_scene_polygon(scene))
    o = OracleCrowdEnv(BodyModel(world["bm"]), world["prior_sd"], _vposer_sd(float(g[pre + "vposer_gain"])), world["mk"], world["feet"],
                       world["fmi"], scene_kind="sdf", sdf_dict=_sdf_tensors(scene), edges=edges, finetuning=bool(g[pre + "finetuning"]))
    c = world["cfg"]
    for k in ("reproj_factor",):
        assert o.cfg[k] == c["modelconfig"][k]
    for k in ("goal_thresh", "max_depth"):
        assert o.cfg[k] == c["trainconfig"][k]
    for k in ("weight_skate", "weight_floor", "weight_face_target", "weight_look_target", "weight_success", "weight_target_dist", "weight_vp"):
        assert o.cfg[k] == c["lossconfig"][k], k
    # ---- sampler (environments.py:65-335) ----
    poses, trans, betas = _motion_seed()
    pair = torch.as_tensor(g[pre + "pair"])
    tr, go, bp, wpath = o.next_body(pair[0:1], pair[1:2], poses, trans, betas)
    _close(tr[0], g[pre + "motion_transl"], "m", pre + "sampler transl")
    _aa_close(go[0], g[pre + "motion_glorot"], pre + "sampler glorot")
    _close(bp[0], g[pre + "motion_body_pose"], "unit", pre + "sampler body_pose")
    _close(betas[0], g[pre + "betas"], "unit", pre + "betas")
    # ---- reset (crowd_env_2f.py:320-415) ----
    obs, accept = o.reset_from(tr, go, bp, betas, wpath)
    assert bool(accept[0]), "the reference's loop accepted this start"
    _close(o.wpath[0], g[pre + "reset_wpath"], "m", pre + "wpath")
    _close(o.state[0], g[pre + "reset_state"], "unit", pre + "reset state")
    _close(o.body_param_seed[0][:, :3], g[pre + "reset_seed"][:, :3], "m", pre + "reset seed transl")
    _aa_close(o.body_param_seed[0][:, 3:6], g[pre + "reset_seed"][:, 3:6], pre + "reset seed glorot")
    _close(o.body_param_seed[0][:, 6:], g[pre + "reset_seed"][:, 6:], "unit", pre + "reset seed pose")
    _close(o.R0[0], g[pre + "reset_R0"], "unit", pre + "reset R0")
    _close(o.T0[0].reshape(-1), g[pre + "reset_T0"].reshape(-1), "m", pre + "reset T0")
    _close(o.dist, g[pre + "reset_dist"], "m", pre + "reset dist")
    _close(obs["state"][0], g[pre + "reset_obs_state"], "unit", pre + "reset obs state")
    _close(obs["egosensing"][0], g[pre + "reset_obs_ego"], "ego", pre + "reset egosensing")
    _close(obs["dist"].reshape(-1), g[pre + "reset_obs_dist"], "unit", pre + "reset obs dist")
    _close(obs["time"].reshape(-1), g[pre + "reset_obs_time"], "unit", pre + "reset obs time")
    # ---- the state manipulations of the generator ----
    if pre + "obstacle_lo" in g:
        o.sdf_dict = _sdf_tensors(synth.make_sdf_scene(world["res"], obstacle=(g[pre + "obstacle_lo"].astype(np.float64),
                                                                              g[pre + "obstacle_hi"].astype(np.float64))))
    if pre + "steps_before" in g:
        o.steps = torch.full((1,), int(g[pre + "steps_before"]), dtype=torch.long)
    if pre + "wpath_override" in g:
        o.wpath = torch.as_tensor(g[pre + "wpath_override"])[None].clone()
    # ---- steps (crowd_env_2f.py:78-317) ----
    for i in range(int(g[pre + "n_steps"])):
        sp = f"{pre}s{i}_"
        z = torch.as_tensor(g[pre + "z"][i])[None]
        obs, rew, term = o.step(z)
        L = o.last
        got = {"Y_gen": L["Y_gen"][:, 0], "pred_params": L["pred_params"][0], "joints": L["joints"][0], "marker_b": L["marker_b"][0],
               "r_skate": L["r_skate"][0], "r_floor": L["r_floor"][0], "r_face_target": L["r_face"][0], "r_look_target": L["r_look"][0],
               "r_goal": L["r_goal"][0], "r_target_dist": L["r_target_dist"][0], "r_vp": L["r_vp"][0], "r_pene": L["r_pene"][0],
               "reward": rew[0], "terminated": term[0], "after_state": o.state[0], "after_seed": o.body_param_seed[0], "after_R0": o.R0[0],
               "after_T0": o.T0[0], "after_dist": o.dist, "obs_ego": obs["egosensing"][0], "obs_dist": obs["dist"], "obs_time": obs["time"]}
        _check_step_common(g, sp, got, counts=L["pene_count"][0].numpy(), w_pene=0.1 if bool(g[pre + "finetuning"]) else 1.0)
        assert bool(L["penetration"][0]) == bool(g[sp + "penetration"])
        _close(np.float64(L["vp_norm"][0]), g[sp + "vp_norm"], "unit", sp + "vp_norm")
        assert int(o.steps[0]) == int(g[sp + "after_steps"])
    # what each case is there for
    last = f"{pre}s{int(g[pre + 'n_steps']) - 1}_"
    if case == "pene_ft":
        assert int(g[last + "num_inside_max"]) >= 40 and bool(g[last + "terminated"])
    if case == "pene":
        assert int(g[f"{pre}s0_num_inside_max"]) >= 40 and not bool(g[f"{pre}s0_terminated"])
    if case == "graze_ft":
        assert 0 < int(g[last + "num_inside_max"]) < 40 and not bool(g[last + "terminated"]) and 0 < float(g[last + "r_pene"]) < 1
    if case == "goal":
        assert float(g[last + "r_goal"]) == 1.0 and bool(g[last + "terminated"])
    if case == "depth":
        assert bool(g[last + "terminated"]) and float(g[last + "obs_time"][0]) == 0.0
    if case == "vposer":
        assert float(g[last + "vp_norm"]) > 11 and float(g[last + "r_vp"]) == 0.0


# ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu_world(world):
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG
    h = BodyModelHandle(world["bm"], world["mk"], world["feet"])
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    combo.load_state_dict(world["prior_sd"])
    combo.cuda().eval()
    return dict(world, handle=h, combo=combo)


@pytest.mark.gpu
@pytest.mark.parametrize("blend", [3, 2, 1, 0])
@pytest.mark.parametrize("case", SDF_CASES)
def test_hip_env_matches_reference_execution(gpu_world, case, blend):
    """VecCrowdEnv (one agent) through the C ABI against the reference's recorded reset / steps, in every blend mode of the
    fused LBS kernel (3 = the default mixed mode: positions as in mode 2, the penetration count's vertices classified through one
    fp16 product for the pose correctives and re-evaluated in fp32 where that cannot decide - its counts are held to the same
    2e-5 m band of the fixture as the other modes; 2 = two-plane bf16 split, 1 = three planes, 0 = fp32 MFMA)."""
    from egogen_amd import _lib, synth
    from egogen_amd.body_model import SdfScene
    from egogen_amd.crowd_env import VecCrowdEnv
    from egogen_amd.models import VPoserEncoder
    w = gpu_world
    g = w["g"]
    pre = case + "_"
    lib = _lib.load()
    old = int(lib.egx_lbs_get_blend_mode())
    _lib.check(lib.egx_lbs_set_blend_mode(blend), "egx_lbs_set_blend_mode")
    try:
        scene = synth.make_sdf_scene(w["res"])
        rings = synth.sdf_scene_polygon(scene)
        vp = VPoserEncoder()
        vsd = seeded_vposer_state_dict()
        vsd.update({k: v for k, v in _vposer_sd(float(g[pre + "vposer_gain"])).items() if k.startswith("bodyprior_enc_mu.")})
        vp.load_state_dict(vsd)
        vp.cuda().eval()
        pair = np.asarray(g[pre + "pair"], np.float32).reshape(1, 2, 3)
        env = VecCrowdEnv(1, w["handle"], w["combo"], vp, finetuning=bool(g[pre + "finetuning"]), seed=0, scene_kind="sdf", sdf_dict=scene,
                          rings=rings, pairs=pair)
        assert env.variant_starts[0] == 5      # environments.py:185 start_frame = 5
        env.set_candidates(pair.reshape(1, 1, 2, 3))
        obs = env.reset()
        assert bool(env.pair_valid_mask[0])
        _close(env.wpath[0], g[pre + "reset_wpath"], "m", pre + "wpath")
        _close(env.state[0], g[pre + "reset_state"], "unit", pre + "reset state")
        _close(env.seed[0][:, :3], g[pre + "reset_seed"][:, :3], "m", pre + "reset seed transl")
        _aa_close(env.seed[0][:, 3:6].cpu(), g[pre + "reset_seed"][:, 3:6], pre + "reset seed glorot")
        _close(env.seed[0][:, 6:], g[pre + "reset_seed"][:, 6:], "unit", pre + "reset seed pose")
        _close(env.R0[0], g[pre + "reset_R0"], "unit", pre + "reset R0")
        _close(env.T0[0], g[pre + "reset_T0"].reshape(-1), "m", pre + "reset T0")
        _close(env.dist, g[pre + "reset_dist"], "m", pre + "reset dist")
        _close(obs["egosensing"][0], g[pre + "reset_obs_ego"], "ego", pre + "reset egosensing")
        _close(obs["dist"].reshape(-1), g[pre + "reset_obs_dist"], "unit", pre + "reset obs dist")
        _close(obs["time"].reshape(-1), g[pre + "reset_obs_time"], "unit", pre + "reset obs time")
        if pre + "obstacle_lo" in g:
            env.sdf = SdfScene(synth.make_sdf_scene(w["res"], obstacle=(g[pre + "obstacle_lo"].astype(np.float64),
                                                                        g[pre + "obstacle_hi"].astype(np.float64))), device=env.dev)
        if pre + "steps_before" in g:
            env.steps.fill_(int(g[pre + "steps_before"]))
        if pre + "wpath_override" in g:
            env.wpath.copy_(torch.as_tensor(g[pre + "wpath_override"])[None])
        for i in range(int(g[pre + "n_steps"])):
            sp = f"{pre}s{i}_"
            z = torch.as_tensor(g[pre + "z"][i])[None].cuda().contiguous()
            obs, rew, term = env.step(z, auto_reset=False)
            rt = env.rterms[0].cpu().numpy()
            got = {"Y_gen": env.Y_gen.reshape(-1, 201),
                   "pred_params": env.pred_params.reshape(20, 93), "joints": env.joints.reshape(20, -1, 3),
                   "r_skate": rt[0], "r_floor": rt[1], "r_face_target": rt[2], "r_look_target": rt[3], "r_goal": rt[4], "r_target_dist": rt[5],
                   "r_pene": rt[6], "r_vp": rt[7], "reward": rew[0], "terminated": term[0], "after_state": env.state[0],
                   "after_seed": env.seed[0].cpu(), "after_R0": env.R0[0], "after_T0": env.T0[0].cpu(), "after_dist": env.dist.cpu(),
                   "obs_ego": obs["egosensing"][0], "obs_dist": obs["dist"].cpu(), "obs_time": obs["time"].cpu()}
            _check_step_common(g, sp, got, counts=env.pene_count.reshape(20).cpu().numpy(), w_pene=0.1 if bool(g[pre + "finetuning"]) else 1.0,
                               band_key="pene_near_zero")
            assert int(env.steps[0]) == int(g[sp + "after_steps"])
    finally:
        _lib.check(lib.egx_lbs_set_blend_mode(old), "egx_lbs_set_blend_mode")


# ----------------------------------------------------------------------------------------------------------------------
# box env (crowd_env_2f_box.py) - env_box_ref.npz
# ----------------------------------------------------------------------------------------------------------------------
BOX_CASES = ["free", "reject", "cover", "graze"]


@pytest.fixture(scope="module")
def box_world():
    from egogen_amd import synth
    g = load_golden("env_box_ref.npz")
    scenes = [synth.box_scene_from_hole(g[f"scene_{k}_box_lo"], g[f"scene_{k}_box_hi"]) for k in ("s0", "s1")]
    for sc, k in zip(scenes, ("s0", "s1")):      # the generator's scenes are reproduced from their holes
        assert np.array_equal(sc["tris"], g[f"scene_{k}_tris"]) and np.array_equal(sc["edges"], g[f"scene_{k}_edges"])
    return {"g": g, "bm": synth.make_body_model(0), "mk": synth.marker_ids(), "feet": synth.feet_vids(), "fmi": synth.feet_marker_idx(),
            "prior_sd": seeded_prior_state_dict(int(g["prior_seed"]), *[float(v) for v in g["prior_gains"]]), "scenes": scenes,
            "cfg": json.loads(str(g["cfg_json"]))}


def _box_scene_list(w, pre):
    from egogen_amd import synth
    g = w["g"]
    scenes = list(w["scenes"])
    if pre + "hole_lo" in g:
        scenes.append(synth.box_scene_from_hole(g[pre + "hole_lo"].astype(np.float64), g[pre + "hole_hi"].astype(np.float64)))
    return scenes


def _box_got(L, rew, term, o, obs):
    return {"Y_gen": L["Y_gen"][:, 0], "pred_params": L["pred_params"][0], "joints": L["joints"][0], "marker_b": L["marker_b"][0],
            "r_skate": L["r_skate"][0], "r_floor": L["r_floor"][0], "r_face_target": L["r_face"][0], "r_look_target": L["r_look"][0],
            "r_goal": L["r_goal"][0], "r_target_dist": L["r_target_dist"][0], "r_vp": L["r_vp"][0], "r_pene": L["r_pene"][0],
            "reward": rew[0], "terminated": term[0], "after_state": o.state[0], "after_seed": o.body_param_seed[0], "after_R0": o.R0[0],
            "after_T0": o.T0[0], "after_dist": o.dist, "obs_ego": obs["egosensing"][0], "obs_dist": obs["dist"], "obs_time": obs["time"]}


@pytest.mark.parametrize("case", BOX_CASES)
def test_oracle_box_env_matches_reference_execution(box_world, case):
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    w = box_world
    g = w["g"]
    pre = case + "_"
    scenes = _box_scene_list(w, pre)
    o = OracleCrowdEnv(BodyModel(w["bm"]), w["prior_sd"], _vposer_sd(1.0), w["mk"], w["feet"], w["fmi"], scene_kind="box", box_scenes=scenes)
    c = w["cfg"]
    assert o.cfg["weight_pene"] == c["lossconfig"]["weight_pene"] and o.cfg["pene_type"] == c["lossconfig"]["pene_type"]
    assert o.cfg["pene_thres"] == c["trainconfig"]["pene_thres"] and o.cfg["max_depth"] == c["trainconfig"]["max_depth"]
    assert o.cfg["map_res"] == c["modelconfig"]["map_res"] and o.cfg["map_extent"] == c["modelconfig"]["map_extent"]
    n = int(g[pre + "n_draws"])
    for i in range(n):      # the reference's `while True` loop: every draw but the last was rejected by the start check
        d = f"{pre}draw{i}_"
        poses, trans, betas = _motion_seed(int(g[d + "start_frame"]))
        pair = torch.as_tensor(g[d + "pair"])
        tr, go, bp, wpath = o.next_body(pair[0:1], pair[1:2], poses, trans, betas, yaw_jitter=torch.tensor([float(g[d + "yaw_jitter"])]))
        _close(tr[0], g[d + "transl"], "m", d + "sampler transl")
        _aa_close(go[0], g[d + "glorot"], d + "sampler glorot")
        _close(wpath[0], g[d + "wpath"], "m", d + "sampler wpath")
        sidx = {"s0": 0, "s1": 1}[str(g[d + "scene"])]
        obs, accept = o.reset_from(tr, go, bp, betas, wpath, scene_idx=[sidx])
        assert bool(accept[0]) == (i == n - 1), (case, i, bool(accept[0]))
    _close(o.state[0], g[pre + "reset_state"], "unit", pre + "reset state")
    _close(o.R0[0], g[pre + "reset_R0"], "unit", pre + "reset R0")
    _close(o.T0[0].reshape(-1), g[pre + "reset_T0"].reshape(-1), "m", pre + "reset T0")
    _close(o.dist, g[pre + "reset_dist"], "m", pre + "reset dist")
    _close(obs["egosensing"][0], g[pre + "reset_obs_ego"], "ego", pre + "reset egosensing")
    _close(obs["dist"].reshape(-1), g[pre + "reset_obs_dist"], "unit", pre + "reset obs dist")
    if pre + "hole_lo" in g:
        o.scene_idx = torch.tensor([2])
    for i in range(int(g[pre + "n_steps"])):
        sp = f"{pre}s{i}_"
        obs, rew, term = o.step(torch.as_tensor(g[pre + "z"][i])[None])
        _check_step_common(g, sp, _box_got(o.last, rew, term, o, obs))
        assert float(o.last["num_pene"][0]) == float(g[sp + "num_pene"])
        assert bool(o.last["penetration"][0]) == bool(g[sp + "penetration"])
    last = f"{pre}s{int(g[pre + 'n_steps']) - 1}_"
    if case == "reject":
        assert n == 3
    if case == "cover":
        assert float(g[f"{pre}s0_num_pene"]) > 3 and bool(g[f"{pre}s0_terminated"]) and float(g[f"{pre}s0_r_pene"]) == 0.0
    if case == "graze":
        assert 0 < float(g[last + "num_pene"]) <= 3 and not bool(g[last + "terminated"]) and float(g[last + "r_pene"]) == pytest.approx(0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("case", BOX_CASES)
def test_hip_box_env_matches_reference_execution(box_world, case):
    """VecCrowdEnv(scene_kind='box') through the C ABI against the reference's recorded draws / reset / steps: the reset kernel
    must commit the draw the reference's loop accepted (`choice`), the step kernels the recorded quantities."""
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.crowd_env import VecCrowdEnv
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder
    w = box_world
    g = w["g"]
    pre = case + "_"
    scenes = _box_scene_list(w, pre)
    h = BodyModelHandle(w["bm"], w["mk"], w["feet"])
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    combo.load_state_dict(w["prior_sd"])
    combo.cuda().eval()
    vp = VPoserEncoder()
    vp.load_state_dict(seeded_vposer_state_dict())
    vp.cuda().eval()
    env = VecCrowdEnv(1, h, combo, vp, seed=0, scene_kind="box", box_scenes=scenes)
    K, n = env.K, int(g[pre + "n_draws"])
    assert n <= K
    idx = [min(i, n - 1) for i in range(K)]          # the recorded draws, the accepted one repeated up to K
    pairs = np.stack([g[f"{pre}draw{i}_pair"] for i in idx]).reshape(1, K, 2, 3)
    yaw = np.array([[float(g[f"{pre}draw{i}_yaw_jitter"]) for i in idx]], np.float32)
    variant = np.array([[env.variant_starts.index(int(g[f"{pre}draw{i}_start_frame"])) for i in idx]])
    scene = np.array([[{"s0": 0, "s1": 1}[str(g[f"{pre}draw{i}_scene"])] for i in idx]])
    env.set_candidates(pairs, yaw, variant, scene)
    obs = env.reset()
    assert int(env.choice[0]) == n - 1 and env.forced_accepts() == 0
    _close(env.wpath[0], g[pre + "reset_wpath"], "m", pre + "wpath")
    _close(env.state[0], g[pre + "reset_state"], "unit", pre + "reset state")
    _close(env.R0[0], g[pre + "reset_R0"], "unit", pre + "reset R0")
    _close(env.T0[0], g[pre + "reset_T0"].reshape(-1), "m", pre + "reset T0")
    _close(env.dist, g[pre + "reset_dist"], "m", pre + "reset dist")
    _close(obs["egosensing"][0], g[pre + "reset_obs_ego"], "ego", pre + "reset egosensing")
    _close(obs["dist"].reshape(-1), g[pre + "reset_obs_dist"], "unit", pre + "reset obs dist")
    if pre + "hole_lo" in g:
        env.scene_idx.fill_(2)
    for i in range(int(g[pre + "n_steps"])):
        sp = f"{pre}s{i}_"
        z = torch.as_tensor(g[pre + "z"][i])[None].cuda().contiguous()
        obs, rew, term = env.step(z, auto_reset=False)
        rt = env.rterms[0].cpu().numpy()
        got = {"Y_gen": env.Y_gen.reshape(-1, 201), "pred_params": env.pred_params.reshape(20, 93), "joints": env.joints.reshape(20, -1, 3),
               "r_skate": rt[0], "r_floor": rt[1], "r_face_target": rt[2], "r_look_target": rt[3], "r_goal": rt[4], "r_target_dist": rt[5],
               "r_pene": rt[6], "r_vp": rt[7], "reward": rew[0], "terminated": term[0], "after_state": env.state[0],
               "after_seed": env.seed[0].cpu(), "after_R0": env.R0[0], "after_T0": env.T0[0].cpu(), "after_dist": env.dist.cpu(),
               "obs_ego": obs["egosensing"][0], "obs_dist": obs["dist"].cpu(), "obs_time": obs["time"].cpu()}
        _check_step_common(g, sp, got)


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE config 5's plumbing (main_crowd_eval.py): tests/golden/env_crowd_ref.npz = four crowd_env_crowd_eval.CrowdEnv members
# built from CrowdMotion.next_body and driven through the reference's OWN DummyCrowdVectorEnv (scripts/gen_env_goldens.py crowd):
# constructor boxes, holes, reset observations, and per round and member the holes it saw, the walkability map, the step's
# quantities and the box it published.
CROWD_CASES = ["ring", "tight"]


@pytest.fixture(scope="module")
def crowd_world():
    from egogen_amd import synth
    g = load_golden("env_crowd_ref.npz")
    return {"g": g, "bm": synth.make_body_model(0), "mk": synth.marker_ids(), "feet": synth.feet_vids(), "fmi": synth.feet_marker_idx(),
            "prior_sd": seeded_prior_state_dict(int(g["prior_seed"]), *[float(v) for v in g["prior_gains"]]),
            "cfg": json.loads(str(g["cfg_json"])), "G": int(g["G"])}


def _ring_box(ring):
    """the reference's 5-point box ring(s) [..., 5, 2] -> (minx, miny, maxx, maxy)"""
    r = np.asarray(ring, np.float64)
    return np.concatenate([r.min(-2), r.max(-2)], -1)


def _others(boxes, k):
    return np.stack([boxes[j] for j in range(len(boxes)) if j != k])[None]      # [S=1, G-1, 4]


@pytest.mark.parametrize("case", CROWD_CASES)
def test_oracle_crowd_env_matches_reference_execution(crowd_world, case):
    """Four oracle members run their OWN state from the recorded sampler inputs; each sees the others' boxes as they stand when it
    steps (members 0..k-1 of this round, k+1.. of the previous one) - the ordering the reference's vector env produced."""
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    w = crowd_world
    g, G = w["g"], w["G"]
    pre = case + "_"
    st = g[pre + "start_target"]                                       # [G,2,3]
    ms, boxes = [], np.zeros((G, 4))
    for k in range(G):
        o = OracleCrowdEnv(BodyModel(w["bm"]), w["prior_sd"], _vposer_sd(1.0), w["mk"], w["feet"], w["fmi"], scene_kind="crowd")
        c = w["cfg"]
        assert o.cfg["pene_type"] == c["lossconfig"]["pene_type"] and o.cfg["pene_thres"] == c["trainconfig"]["pene_thres"]
        assert o.cfg["max_depth"] == c["trainconfig"]["max_depth"] and o.cfg["map_res"] == c["modelconfig"]["map_res"]
        mp = f"{pre}m{k}_"
        poses, trans, betas = _motion_seed(int(g[mp + "start_frame"]))
        tr, go, bp, wpath = o.next_body(torch.as_tensor(st[k, 0:1]), torch.as_tensor(st[k, 1:2]), poses, trans, betas,
                                        yaw_jitter=torch.tensor([float(g[mp + "yaw_jitter"])]))
        _close(tr[0], g[mp + "transl"], "m", mp + "sampler transl")
        _aa_close(go[0], g[mp + "glorot"], mp + "sampler glorot")
        _close(wpath[0], g[mp + "wpath"], "m", mp + "sampler wpath")
        o.set_crowd_boxes(np.zeros((1, G - 1, 4)))
        o.reset_from(tr, go, bp, betas, wpath)
        boxes[k] = o.own_bbox().numpy()[0]
        _close(boxes[k], _ring_box(g[mp + "init_bbox"]), "m", mp + "constructor box")
        ms.append((o, (tr, go, bp, betas, wpath)))
    for k, (o, args) in enumerate(ms):       # DummyCrowdVectorEnv.__init__ hands every member the others' boxes, THEN reset()
        mp = f"{pre}m{k}_"
        _close(_others(boxes, k)[0], _ring_box(g[mp + "init_holes"]), "m", mp + "initial holes")
        o.set_crowd_boxes(_others(boxes, k))
        obs, _ = o.reset_from(*args)
        _close(o.state[0], g[mp + "reset_state"], "unit", mp + "reset state")
        _close(o.R0[0], g[mp + "reset_R0"], "unit", mp + "reset R0")
        _close(o.T0[0].reshape(-1), g[mp + "reset_T0"].reshape(-1), "m", mp + "reset T0")
        _close(o.dist, g[mp + "reset_dist"], "m", mp + "reset dist")
        _close(obs["egosensing"][0], g[mp + "reset_obs_ego"], "ego", mp + "reset egosensing")
        _close(obs["dist"].reshape(-1), g[mp + "reset_obs_dist"], "unit", mp + "reset obs dist")
    saw_cells = 0
    for r in range(int(g[pre + "n_rounds"])):
        for k, (o, _) in enumerate(ms):
            sp = f"{pre}r{r}_m{k}_"
            _close(_others(boxes, k)[0], _ring_box(g[sp + "holes_seen"]), "m", sp + "holes at step time")
            o.set_crowd_boxes(_others(boxes, k))
            obs, rew, term = o.step(torch.as_tensor(g[pre + "z"][r, k])[None])
            _check_step_common(g, sp, _box_got(o.last, rew, term, o, obs))
            assert np.array_equal(o.last["local_map"][0].numpy(), g[sp + "local_map"]), sp + "walkability map"
            assert float(o.last["num_pene"][0]) == float(g[sp + "num_pene"])
            assert bool(o.last["penetration"][0]) == bool(g[sp + "penetration"])
            boxes[k] = o.own_bbox().numpy()[0]
            _close(boxes[k], _ring_box(g[sp + "bbox_after"]), "m", sp + "published box")
            saw_cells += int((g[sp + "local_map"] < 0).sum())
    assert saw_cells > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", CROWD_CASES)
def test_hip_crowd_group_matches_reference_execution(crowd_world, case):
    """CrowdGroupEnv (the product's DummyCrowdVectorEnv + crowd_env_crowd_eval.CrowdEnv, S = 1 scene of G = 4 members) through the
    C ABI against the same recording: reset sequence, then `step` of the whole group per round."""
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.crowd_env import CrowdGroupEnv
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder
    w = crowd_world
    g, G = w["g"], w["G"]
    pre = case + "_"
    h = BodyModelHandle(w["bm"], w["mk"], w["feet"])
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    combo.load_state_dict(w["prior_sd"])
    combo.cuda().eval()
    vp = VPoserEncoder()
    vp.load_state_dict(seeded_vposer_state_dict())
    vp.cuda().eval()
    st = g[pre + "start_target"].reshape(G, 1, 2, 3)
    grp = CrowdGroupEnv(1, st, h, combo, vp, seed=0)
    for k, m in enumerate(grp.members):
        mp = f"{pre}m{k}_"
        variant = np.array([m.variant_starts.index(int(g[mp + "start_frame"]))])
        m.set_candidates(st[k].reshape(1, 1, 2, 3), np.array([float(g[mp + "yaw_jitter"])], np.float32), variant)
        m._launch_reset(None)                    # publishes the constructor box (CrowdGroupEnv.reset's first pass)
    for k, m in enumerate(grp.members):
        mp = f"{pre}m{k}_"
        _close(grp.bbox[k, 0], _ring_box(g[mp + "init_bbox"]), "m", mp + "constructor box")
    for k, m in enumerate(grp.members):
        mp = f"{pre}m{k}_"
        m._launch_reset(None)                    # second pass: the observation sees every member's box
        obs = m.obs()
        _close(m.wpath[0], g[mp + "reset_wpath"], "m", mp + "wpath")
        _close(m.state[0], g[mp + "reset_state"], "unit", mp + "reset state")
        _close(m.R0[0], g[mp + "reset_R0"], "unit", mp + "reset R0")
        _close(m.T0[0], g[mp + "reset_T0"].reshape(-1), "m", mp + "reset T0")
        _close(m.dist, g[mp + "reset_dist"], "m", mp + "reset dist")
        _close(obs["egosensing"][0], g[mp + "reset_obs_ego"], "ego", mp + "reset egosensing")
        _close(obs["dist"].reshape(-1), g[mp + "reset_obs_dist"], "unit", mp + "reset obs dist")
    for r in range(int(g[pre + "n_rounds"])):
        z = torch.as_tensor(g[pre + "z"][r]).cuda()
        # the group's own loop (CrowdGroupEnv.step), member by member so that each member's buffers can be read before the next one
        # overwrites the shared box table
        for k, m in enumerate(grp.members):
            sp = f"{pre}r{r}_m{k}_"
            obs, rew, term = m.step(z[k:k + 1].contiguous(), auto_reset=False)
            rt = m.rterms[0].cpu().numpy()
            got = {"Y_gen": m.Y_gen.reshape(-1, 201), "pred_params": m.pred_params.reshape(20, 93), "joints": m.joints.reshape(20, -1, 3),
                   "r_skate": rt[0], "r_floor": rt[1], "r_face_target": rt[2], "r_look_target": rt[3], "r_goal": rt[4], "r_target_dist": rt[5],
                   "r_pene": rt[6], "r_vp": rt[7], "reward": rew[0], "terminated": term[0], "after_state": m.state[0],
                   "after_seed": m.seed[0].cpu(), "after_R0": m.R0[0], "after_T0": m.T0[0].cpu(), "after_dist": m.dist.cpu(),
                   "obs_ego": obs["egosensing"][0], "obs_dist": obs["dist"].cpu(), "obs_time": obs["time"].cpu()}
            _check_step_common(g, sp, got)
            _close(grp.bbox[k, 0], _ring_box(g[sp + "bbox_after"]), "m", sp + "published box")


# ----------------------------------------------------------------------------------------------------------------------
# vis.py::rollout_primitives (:44-78), executed on a seeded motion list (scripts/gen_env_goldens.py rollout)
def _rollout_fixture():
    g = load_golden("rollout_prims_ref.npz")
    mps = [{"smplx_params": g[f"p{i}_smplx_params"].copy(), "betas": g["betas"], "gender": "male", "transf_rotmat": g[f"p{i}_rotmat"],
            "transf_transl": g[f"p{i}_transl"], "mp_type": str(g[f"p{i}_mp_type"])} for i in range(int(g["n"]))]
    return g, mps


def _check_rollout(got, ref):
    from scipy.spatial.transform import Rotation
    assert got.shape == ref.shape
    _close(got[:, :3], ref[:, :3], "m", "rolled-out transl")
    _close(Rotation.from_rotvec(got[:, 3:6]).as_matrix(), Rotation.from_rotvec(ref[:, 3:6]).as_matrix(), "unit", "rolled-out glorot")
    assert np.array_equal(np.asarray(got[:, 6:], np.float32), np.asarray(ref[:, 6:], np.float32))


def test_oracle_rollout_primitives_matches_reference_execution():
    from egogen_amd import synth
    from oracle.rollout import rollout_primitives
    from oracle.smplx_lbs import BodyModel, smplx_forward
    g, mps = _rollout_fixture()
    ob = BodyModel(synth.make_body_model(int(g["body_model_seed"])))

    def pelvis_of(b):
        _, j = smplx_forward(ob, torch.zeros(1, 93), torch.as_tensor(b).reshape(1, 10).float())
        return j[0, 0].numpy()
    assert g["sequence"].shape == (20 + 18 + 19, 93)        # first primitive whole, a '2-frame' one drops two rows, a '1-frame' one
    _check_rollout(rollout_primitives(mps, pelvis_of), g["sequence"])


@pytest.mark.gpu
def test_hip_rollout_primitives_matches_reference_execution():
    from egogen_amd import synth
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.utils import rollout_primitives
    g, mps = _rollout_fixture()
    h = BodyModelHandle(synth.make_body_model(int(g["body_model_seed"])), synth.marker_ids(), synth.feet_vids())
    _check_rollout(rollout_primitives(mps, h), g["sequence"])


# ----------------------------------------------------------------------------------------------------------------------
# The EgoBody evaluation (main_egobody_eval.py): tests/golden/env_egobody_ref.npz = two crowd_env_egobody_eval.CrowdEnv members built
# from Egobody.gen_init_body inside the walkable region of the in-tree room_0 navmesh, under the reference's DummyCrowdVectorEnv
# (scripts/gen_env_goldens.py egobody).  Cases: two clean rounds; a pose the filter at 14 rejects (`exit(-1)`, :229-234); a pelvis
# that leaves the region within the first five steps (`exit(-1)`, :208-216).  The fixture cannot pin what shapely does with
# `Polygon(polygon, holes)` (:824) - the generator's stand-in ASSUMES it returns the polygon (static scene), like the oracle.
EGO_CASES = ["pair", "pose", "pelvis"]
_EXIT_FLAG = {"invalid pelvis location": 1, "unrealistic pose": 2}


@pytest.fixture(scope="module")
def ego_world():
    from egogen_amd import synth
    g = load_golden("env_egobody_ref.npz")
    rings = [g[f"ring{i}"] for i in range(int(g["n_rings"]))]
    return {"g": g, "bm": synth.make_body_model(0), "mk": synth.marker_ids(), "feet": synth.feet_vids(), "fmi": synth.feet_marker_idx(),
            "prior_sd": seeded_prior_state_dict(int(g["prior_seed"]), *[float(v) for v in g["prior_gains"]]), "rings": rings}


def _ego_seed(g, mp):
    poses, trans, _ = _motion_seed(int(g[mp + "start_frame"]))
    return poses, trans, torch.as_tensor(g[mp + "betas"], dtype=torch.float32).reshape(1, 10)


@pytest.mark.parametrize("case", EGO_CASES)
def test_oracle_egobody_env_matches_reference_execution(ego_world, case):
    from egogen_amd import synth
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    w = ego_world
    g, G = w["g"], 2
    pre = case + "_"
    st = g[pre + "start_target"]
    edges = synth.rings_to_edges(w["rings"])
    ms, boxes = [], np.zeros((G, 4))
    for k in range(G):
        o = OracleCrowdEnv(BodyModel(w["bm"]), w["prior_sd"], _vposer_sd(float(g[pre + "vposer_gain"])), w["mk"], w["feet"], w["fmi"],
                           scene_kind="crowd")
        o.set_egobody(edges, static=True, vp_thresh=14.0)
        mp = f"{pre}m{k}_"
        poses, trans, betas = _ego_seed(g, mp)
        tr, go, bp, wpath = o.next_body(torch.as_tensor(st[k, 0:1]), torch.as_tensor(st[k, 1:2]), poses, trans, betas,
                                        yaw_jitter=torch.tensor([float(g[mp + "yaw_jitter"])]))
        _close(tr[0], g[mp + "transl"], "m", mp + "sampler transl")
        _aa_close(go[0], g[mp + "glorot"], mp + "sampler glorot")
        _close(wpath[0], g[mp + "wpath"], "m", mp + "sampler wpath")
        o.set_crowd_boxes(np.zeros((1, G - 1, 4)))
        obs, _ = o.reset_from(tr, go, bp, betas, wpath)
        boxes[k] = o.own_bbox().numpy()[0]
        _close(boxes[k], _ring_box(g[mp + "init_bbox"]), "m", mp + "constructor box")
        _close(o.state[0], g[mp + "reset_state"], "unit", mp + "reset state")
        _close(o.T0[0].reshape(-1), g[mp + "reset_T0"].reshape(-1), "m", mp + "reset T0")
        _close(obs["egosensing"][0], g[mp + "reset_obs_ego"], "ego", mp + "reset egosensing")     # static scene: nobody else in it
        _close(obs["dist"].reshape(-1), g[mp + "reset_obs_dist"], "unit", mp + "reset obs dist")
        ms.append(o)
    ex_r, ex_k = int(g[pre + "exit_round"]), int(g[pre + "exit_member"])
    n_rounds = ex_r + 1 if ex_r >= 0 else int(g[pre + "n_rounds"])
    for r in range(n_rounds):
        for k, o in enumerate(ms):
            sp = f"{pre}r{r}_m{k}_"
            o.set_crowd_boxes(_others(boxes, k))
            obs, rew, term = o.step(torch.as_tensor(g[pre + "z"][r, k])[None])
            if (r, k) == (ex_r, ex_k):      # the reference left the process here
                assert int(o.last["invalid"][0]) & _EXIT_FLAG[str(g[pre + "exit_reason"])], (sp, int(o.last["invalid"][0]))
                return
            assert int(o.last["invalid"][0]) == 0, sp
            _close(_others(boxes, k)[0], _ring_box(g[sp + "holes_seen"]), "m", sp + "holes at step time")
            _check_step_common(g, sp, _box_got(o.last, rew, term, o, obs))
            assert np.array_equal(o.last["local_map"][0].numpy(), g[sp + "local_map"]), sp + "walkability map"
            assert float(o.last["num_pene"][0]) == float(g[sp + "num_pene"]) and not bool(g[sp + "terminated"])
            boxes[k] = o.own_bbox().numpy()[0]
            _close(boxes[k], _ring_box(g[sp + "bbox_after"]), "m", sp + "published box")
    assert ex_r < 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", EGO_CASES)
def test_hip_egobody_group_matches_reference_execution(ego_world, case):
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.crowd_env import CrowdGroupEnv
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder
    w = ego_world
    g, G = w["g"], 2
    pre = case + "_"
    h = BodyModelHandle(w["bm"], w["mk"], w["feet"])
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    combo.load_state_dict(w["prior_sd"])
    combo.cuda().eval()
    vp = VPoserEncoder()
    vp.load_state_dict({k: v for k, v in _vposer_sd(float(g[pre + "vposer_gain"])).items()})
    vp.cuda().eval()
    st = g[pre + "start_target"].reshape(G, 1, 2, 3)
    seeds = []
    for k in range(G):
        poses, trans, betas = _ego_seed(g, f"{pre}m{k}_")
        seeds.append([{"poses": poses[0].numpy().astype(np.float64), "trans": trans[0].numpy().astype(np.float64),
                       "betas": betas[0].numpy().astype(np.float64)}])
    grp = CrowdGroupEnv(1, st, h, combo, vp, seed=0, scene_rings=w["rings"], static_scene=True, agent_seeds=seeds, vp_thresh=14.0,
                        goal_terminates=False)
    for k, m in enumerate(grp.members):
        m.set_candidates(st[k].reshape(1, 1, 2, 3), np.array([float(g[f"{pre}m{k}_yaw_jitter"])], np.float32), np.arange(1))
        m._launch_reset(None)
    for k, m in enumerate(grp.members):
        mp = f"{pre}m{k}_"
        m._launch_reset(None)
        obs = m.obs()
        _close(grp.bbox[k, 0], _ring_box(g[mp + "init_bbox"]), "m", mp + "constructor box")
        _close(m.wpath[0], g[mp + "reset_wpath"], "m", mp + "wpath")
        _close(m.state[0], g[mp + "reset_state"], "unit", mp + "reset state")
        _close(m.T0[0], g[mp + "reset_T0"].reshape(-1), "m", mp + "reset T0")
        _close(obs["egosensing"][0], g[mp + "reset_obs_ego"], "ego", mp + "reset egosensing")
    ex_r, ex_k = int(g[pre + "exit_round"]), int(g[pre + "exit_member"])
    n_rounds = ex_r + 1 if ex_r >= 0 else int(g[pre + "n_rounds"])
    for r in range(n_rounds):
        z = torch.as_tensor(g[pre + "z"][r]).cuda()
        for k, m in enumerate(grp.members):
            sp = f"{pre}r{r}_m{k}_"
            obs, rew, term = m.step(z[k:k + 1].contiguous(), auto_reset=False)
            if (r, k) == (ex_r, ex_k):
                assert int(m.invalid[0]) & _EXIT_FLAG[str(g[pre + "exit_reason"])], (sp, int(m.invalid[0]))
                return
            assert int(m.invalid[0]) == 0, sp
            rt = m.rterms[0].cpu().numpy()
            got = {"Y_gen": m.Y_gen.reshape(-1, 201), "pred_params": m.pred_params.reshape(20, 93), "joints": m.joints.reshape(20, -1, 3),
                   "r_skate": rt[0], "r_floor": rt[1], "r_face_target": rt[2], "r_look_target": rt[3], "r_goal": rt[4], "r_target_dist": rt[5],
                   "r_pene": rt[6], "r_vp": rt[7], "reward": rew[0], "terminated": term[0], "after_state": m.state[0],
                   "after_seed": m.seed[0].cpu(), "after_R0": m.R0[0], "after_T0": m.T0[0].cpu(), "after_dist": m.dist.cpu(),
                   "obs_ego": obs["egosensing"][0], "obs_dist": obs["dist"].cpu(), "obs_time": obs["time"].cpu()}
            _check_step_common(g, sp, got)
            _close(grp.bbox[k, 0], _ring_box(g[sp + "bbox_after"]), "m", sp + "published box")
    assert ex_r < 0
