#!/usr/bin/env python3
"""Golden vectors of the ENVIRONMENT COMPOSITE and of the PPO update, produced by EXECUTING the reference's own code
(build container only - /root/reference does not exist on the GPU box):

  crowd_ppo/crowd_env_2f.py::CrowdEnv.reset (:320-415) / .step (:78-317)            -> env_step_ref.npz   (`sdf`)
  crowd_ppo/crowd_env_2f_box.py::CrowdEnv.reset (:342-431) / .step (:78-340)        -> env_box_ref.npz    (`box`)
  exp_GAMMAPrimitive/utils/environments.py::BatchGeneratorScene2frameTrain.next_body (:65-335) and
      BatchGeneratorScene2frameTrainBox.next_body (:371-627)                         (inside the two resets)
  models/baseops.py::SMPLXParser (forward_smplx / get_new_coordinate / update_transl_glorot / calc_calibrate_offset)
  models/models_GAMMA_primitive.py::GAMMAPrimitiveCombo.sample_prior                (the reference classes, seeded weights)
  crowd_ppo/ppo_policy.py::GAMMAPPOPolicy.learn (:182-265)                          -> ppo_learn_ref.npz  (`learn`)
  crowd_ppo/crowd_env_crowd_eval.py::CrowdEnv (:44-454, :742-837) x 4 under crowd_ppo/dummy_vector_env.py::DummyCrowdVectorEnv,
      bodies from environments.py::CrowdMotion.next_body (:1041-1157)               -> env_crowd_ref.npz  (`crowd`)
  vis.py::rollout_primitives (:44-78)                                                -> rollout_prims_ref.npz (`rollout`)
  crowd_ppo/crowd_env_egobody_eval.py::CrowdEnv x 2 under DummyCrowdVectorEnv, bodies from environments.py::Egobody.gen_init_body
      (:679-765) in the walkable region of data/room_0's navmesh                    -> env_egobody_ref.npz (`egobody`)

What is substituted, and by what (the packages are absent from this image):
  smplx.create(...)                      -> adapter around oracle/smplx_lbs.py on the synthetic full-size body (V = 10 475)
  torchgeometry.{angle_axis_to_rotation_matrix, rotation_matrix_to_angle_axis}, pytorch3d.transforms.{axis_angle_to_matrix,
      matrix_to_axis_angle, euler_angles_to_matrix} -> oracle/rot.py's restatements of those releases
  human_body_prior VPoser `.encode(x).loc` -> oracle/nets.py::vposer_encode on seeded weights
  CrowdEnv._calc_egosensing (shapely)    -> oracle/env.py::calc_egosensing (the only METHOD of the env that is replaced)
  trimesh.load / the navmesh             -> a (vertices, faces) namespace of the synthetic scene
  tianshou.policy.PPOPolicy / Batch / to_torch_as (learn only) -> a 40-line stand-in holding the attributes `learn` reads
  shapely Polygon / union_all / Point (crowd only) -> RectPolygon: exact for the floor square + box holes that env builds
  tianshou.env.DummyVectorEnv (crowd only) -> a stand-in whose workers step at `send`, like tianshou's DummyEnvWorker
  device spellings ('cuda', torch.cuda.FloatTensor, Tensor.cuda()) -> CPU
The fixtures therefore pin the env / loss COMPOSITE's own arithmetic - reward block and its thresholds (40 vertices, 0.075,
0.02, 11), termination, state / seed / frame bookkeeping, the sampler's rotations, the loss terms and the clip quirk - on top of
pieces that are pinned (nets, calc_sdf, get_map, features) or restated (smplx, rotations, VPoser) elsewhere.

Locals of `step` (the eight reward terms, counts, intermediate tensors) are read from its frame at the moment it calls
`_calc_egosensing` (after the reward block) - nothing of the method is re-typed here.
"""
import json
import os
import pickle
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import gen_goldens as gg  # noqa: E402

REF, OUT = gg.REF, gg.OUT
NB = 4        # n_gens_2frame: the reference replicates every env's batch x4 (crowd_env_2f.py:29)


class AttrDict(dict):
    """omegaconf.DictConfig as the env uses it: attribute and item access on nested dicts."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) else v


class _Out:
    pass


class FakeSMPLX:
    """smplx.SMPLX call signature (keyword tensors; parameters that are not passed are the module's zero parameters of the
    construction batch size) on the oracle LBS."""

    def __init__(self, bm, batch_size):
        self.bm, self.batch_size = bm, batch_size
        self.faces = np.zeros((1, 3), np.int64)

    def eval(self):
        return self

    def cuda(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, return_verts=True, transl=None, global_orient=None, body_pose=None, betas=None, left_hand_pose=None,
                 right_hand_pose=None, **kw):
        from oracle.smplx_lbs import smplx_forward
        given = [t for t in (transl, global_orient, body_pose, betas, left_hand_pose, right_hand_pose) if t is not None]
        n = max([t.shape[0] for t in given] + [1]) if given else self.batch_size
        if all(t.shape[0] == 1 for t in given) and given:
            n = 1
        if not any(t is not None for t in (transl, global_orient, body_pose)):
            n = self.batch_size
        xb = torch.zeros(n, 93)
        for t, sl in ((transl, slice(0, 3)), (global_orient, slice(3, 6)), (body_pose, slice(6, 69)), (left_hand_pose, slice(69, 81)),
                      (right_hand_pose, slice(81, 93))):
            if t is not None:
                xb[:, sl] = t
        b = torch.zeros(n, 10) if betas is None else betas.reshape(-1, 10).expand(n, 10)
        v, j = smplx_forward(self.bm, xb.to(self.bm.dtype), b.to(self.bm.dtype))
        o = _Out()
        o.vertices, o.joints = v.float(), j.float()
        return o


def _euler_xyz(euler, convention="XYZ"):
    """pytorch3d 0.7.4 euler_angles_to_matrix [upstream]: R = R_c0(e0) R_c1(e1) R_c2(e2)."""
    def axis_rot(axis, a):
        c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
        flat = {"X": (o, z, z, z, c, -s, z, s, c), "Y": (c, z, s, z, o, z, -s, z, c), "Z": (c, -s, z, s, c, z, z, z, o)}[axis]
        return torch.stack(flat, -1).reshape(a.shape + (3, 3))
    ms = [axis_rot(c, e) for c, e in zip(convention, torch.unbind(euler, -1))]
    return ms[0] @ ms[1] @ ms[2]


class cpu_world(gg.cuda_to_cpu):
    """gen_goldens.cuda_to_cpu plus Tensor.cuda() / Module.cuda() / torch.cuda.LongTensor / torch.eye(..).cuda()."""

    def __enter__(self):
        super().__enter__()
        self._tc, self._mc, self._lt = torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.LongTensor
        torch.Tensor.cuda = lambda t, *a, **k: t
        torch.nn.Module.cuda = lambda m, *a, **k: m
        torch.cuda.LongTensor = torch.LongTensor
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.LongTensor = self._tc, self._mc, self._lt
        super().__exit__(*exc)


def install(recorder):
    """Stubs + substitutes (module docstring).  `recorder` collects the random draws of the samplers."""
    from oracle import rot as orot
    gg.install_stubs()
    gg.install_env_stubs()
    tgm = sys.modules["torchgeometry"]
    tgm.rotation_matrix_to_angle_axis = lambda m: orot.tgm_rotation_matrix_to_angle_axis(m[:, :3, :3])
    tgm.angle_axis_to_rotation_matrix = lambda aa: orot.tgm_angle_axis_to_rotation_matrix(aa)
    p3t = sys.modules["pytorch3d.transforms"]
    p3t.axis_angle_to_matrix = orot.p3d_axis_angle_to_matrix
    p3t.matrix_to_axis_angle = orot.p3d_matrix_to_axis_angle

    def euler(e, convention="XYZ"):
        recorder.setdefault("euler_z", []).append(float(e.reshape(-1)[2]))
        return _euler_xyz(e, convention)
    p3t.euler_angles_to_matrix = euler
    sys.modules["pytorch3d"].transforms = p3t
    st = sys.modules["pytorch3d.structures"]
    st.Meshes = lambda **k: None
    sys.modules["pytorch3d"].structures = st
    class _Space:                     # gymnasium.spaces.Box / Dict: constructed in CrowdEnv.__init__, never read afterwards
        def __init__(self, *a, **k):
            pass
    sys.modules["gymnasium.spaces"].Box = sys.modules["gymnasium.spaces"].Dict = _Space
    # names the sampler module imports from shapely at module level (never called on the paths exercised)
    sh = sys.modules["shapely"]
    sh.union_all = None
    for n in ("Polygon", "Point", "MultiPoint", "mapping"):
        setattr(sys.modules["shapely.geometry"], n, object)


def reference_first():
    """`crowd_ppo` names a package in this repo AND a directory of the reference: import everything of the repo that the
    adapters need first, then take the repo root off sys.path so that `crowd_ppo.*` resolves under /root/reference/motion."""
    import egogen_amd.synth, oracle.env, oracle.nets, oracle.rot, oracle.smplx_lbs, oracle.ppo, tests.helpers  # noqa: F401,E401
    root = os.path.abspath(os.path.join(HERE, ".."))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != root]
    sys.modules.pop("crowd_ppo", None)
    sys.path.insert(0, REF)


def load_yaml(name):
    import yaml
    with open(os.path.join(REF, "crowd_ppo", "cfg_samp20", name)) as f:
        return yaml.safe_load(f)


PRIOR_SEED, PRIOR_GAINS = 200, (0.25, 0.25)     # a TAME seeded motion prior: the body stays within ~1 m of its start per primitive,
# so the penetration counts live around the thresholds (the suite's default gains 0.7 / 0.6 throw it ~5 m: every count saturates)


def build_combo(mgp):
    """The reference's GAMMAPrimitiveCombo with seeded weights (tests/helpers.py::seeded_prior_state_dict)."""
    from tests.helpers import seeded_prior_state_dict
    combo_cfg = load_yaml("MPVAECombo_samp_2frame.yml")["modelconfig"]
    pcfg = load_yaml(combo_cfg["predictor_config"] + ".yml")["modelconfig"]
    rcfg = load_yaml(combo_cfg["regressor_config"] + ".yml")["modelconfig"]
    combo = mgp.GAMMAPrimitiveCombo(pcfg, rcfg)
    sd = seeded_prior_state_dict(PRIOR_SEED, *PRIOR_GAINS)
    missing, unexpected = combo.load_state_dict(sd, strict=False)
    assert not unexpected and all("markers" in k or "bm" in k for k in missing), (missing, unexpected)
    combo.eval()
    return types.SimpleNamespace(model=combo)


class FakeVPoser:
    def __init__(self, sd):
        self.sd = sd

    def encode(self, body_pose):
        from oracle.nets import vposer_encode
        return types.SimpleNamespace(loc=vposer_encode(self.sd, body_pose))


_STEP_LOCALS = {}


def ego_hook(self, joint):
    """Stands in for CrowdEnv._calc_egosensing (shapely).  `step` calls it after the whole reward block (:296), so the caller's
    frame holds every local of that block: snapshot it (sys.setprofile / settrace cannot be used - SMPLXParser.get_jts builds its
    keyword arguments from locals(), which a trace function re-synchronises)."""
    from oracle.env import calc_egosensing
    fr = sys._getframe(1)
    if fr.f_code.co_name == "step":
        _STEP_LOCALS.clear()
        _STEP_LOCALS.update(fr.f_locals)
    return calc_egosensing(joint, self.scene_poly).float()


def capture_step(env_cls, env, z):
    """Call the reference's `step` and return (its return value, its local variables when it calls _calc_egosensing)."""
    _STEP_LOCALS.clear()
    ret = env.step(z)
    return ret, dict(_STEP_LOCALS)


def t2n(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def record_state(env, pre):
    return {pre + "state": t2n(env.state), pre + "seed": t2n(env.body_param_seed[0]), pre + "R0": t2n(env.R0[0]), pre + "T0": t2n(env.T0[0]),
            pre + "dist": t2n(env.dist).reshape(-1)[:1], pre + "wpath": t2n(env.body_scene_data["wpath"]), pre + "steps": np.int64(env.steps)}


STEP_LOCALS = ("r_skate", "r_floor", "r_face_target", "r_look_target", "r_goal", "r_target_dist", "r_pene", "r_vp", "vp_norm")


def record_step(ret, loc, env, pre, box=False):
    obs, reward, terminated, truncated, _ = ret
    out = {pre + "obs_state": t2n(obs["state"]), pre + "obs_ego": t2n(obs["egosensing"]), pre + "obs_dist": t2n(obs["dist"]).reshape(-1),
           pre + "obs_time": t2n(obs["time"]).reshape(-1), pre + "reward": np.float64(reward), pre + "terminated": np.bool_(terminated),
           pre + "truncated": np.bool_(truncated)}
    for k in STEP_LOCALS:
        out[pre + k] = np.float64(t2n(loc[k]).reshape(-1)[0])
    out[pre + "penetration"] = np.bool_(loc["penetration"])
    if box:
        out[pre + "num_pene"] = np.float64(t2n(loc["num_pene"]).reshape(-1)[0])
    else:
        out[pre + "num_inside_max"] = np.int64(t2n(loc["num_inside_max"]))
        out[pre + "pene_count"] = t2n(loc["sdf_values"].lt(0.0).sum(dim=-1)[0]).astype(np.int64)       # [20] (feet zeroed)
        near = loc["sdf_values"][0].abs() < 2e-5        # vertices within fp32 round-off of the zero level set: their sign is not
        near[:, env.feet_vids] = False                  # reproducible by another fp32 evaluation order (tests bound |d count| by it)
        out[pre + "pene_near_zero"] = t2n(near.sum(-1)).astype(np.int64)
        near6 = loc["sdf_values"][0].abs() < 6e-5       # the band of the library's mixed blend mode (fp16 pose-corrective product on the
        near6[:, env.feet_vids] = False                 # tiles that only feed this count)
        out[pre + "pene_near_6e5"] = t2n(near6.sum(-1)).astype(np.int64)
    out[pre + "Y_gen"] = t2n(loc["Y_gen"][:, 0])                       # [18,201]
    out[pre + "pred_params"] = t2n(loc["pred_params"][0])             # [20,93] (after _blend_params)
    out[pre + "joints"] = t2n(loc["pred_output"].joints.reshape(NB, 20, -1, 3)[0])
    out[pre + "marker_b"] = t2n(loc["pred_marker_b"][0])
    out.update(record_state(env, pre + "after_"))
    return out


def body_and_parsers(baseops, smplx_mod, V=None):
    from egogen_amd import synth
    from oracle.smplx_lbs import BodyModel
    bm = BodyModel(synth.make_body_model(0))
    smplx_mod.create = lambda *a, batch_size=1, **k: FakeSMPLX(bm, batch_size)
    mk = dict(device="cpu", marker_placement="ssm2_67")
    parsers = [baseops.SMPLXParser(dict(mk, n_batch=n * NB)) for n in (1, 2, 20)]
    return bm, parsers


def feet_marker_idx_and_markers():
    with open(os.path.join(REF, "data", "SSM2.json")) as f:
        d = json.load(f)["markersets"][0]["indices"]
    feet = ["RHEE", "RTOE", "RRSTBEEF", "LHEE", "LTOE", "LRSTBEEF"]          # main_ppo.py:296-298
    return [list(d.keys()).index(n) for n in feet], list(d.values())


def sdf_scene_tensors(scene):
    return {k: torch.as_tensor(np.asarray(scene[k]), dtype=torch.float32) for k in ("sdf", "center", "scale")}


# ------------------------------------------------------------------------------------------------------------------
def gen_sdf():
    from egogen_amd import synth
    from oracle.env import calc_egosensing
    from tests.helpers import seeded_vposer_state_dict
    rec = {}
    install(rec)
    cwd = os.getcwd()
    os.chdir(REF)       # the env / parser / sampler open data/*.json relative to motion/
    reference_first()
    try:
        with cpu_world():
            from crowd_ppo import crowd_env_2f as ce
            from exp_GAMMAPrimitive.utils import environments as envs
            from models import baseops, models_GAMMA_primitive as mgp
            cfg = AttrDict(load_yaml("MPVAEPolicy_samp_collision.yaml"))
            bm, (p1, p2, pmp) = body_and_parsers(baseops, sys.modules["smplx"])
            fmi, markers = feet_marker_idx_and_markers()
            assert markers == [int(v) for v in synth.marker_ids()] and fmi == list(synth.feet_marker_idx())
            genop = build_combo(mgp)
            vsd = {k: v.float() for k, v in seeded_vposer_state_dict().items()}
            RES = 64
            scene_free = synth.make_sdf_scene(RES)
            rings = synth.sdf_scene_polygon(scene_free)
            edges = synth.rings_to_edges(rings)
            # the sampler, without its constructor (it loads licensed body models / the Replica navmesh)
            sampler = object.__new__(envs.BatchGeneratorScene2frameTrain)
            sampler.scene_list, sampler.scene_type, sampler.index_rec = None, "room_0", 0
            from pathlib import Path
            sampler.scene_dir = Path("data/room_0")
            sampler.navmesh_path = sampler.scene_dir / "navmesh_tight.ply"
            sampler.navmesh = types.SimpleNamespace(vertices=np.zeros((3, 3)), faces=np.zeros((1, 3), np.int64), visual=types.SimpleNamespace())
            sampler.shapely_poly = edges                      # what `_calc_egosensing` (replaced below) receives as self.scene_poly
            sampler.motion_data = np.load("data/locomotion/subseq_00343.npz")
            sampler.bm_2frame = FakeSMPLX(bm, 2)
            ce.CrowdEnv._calc_egosensing = ego_hook
            out = {"sdf_res": np.int64(RES), "body_model_seed": np.int64(0), "n_cases": np.int64(0), "prior_seed": np.int64(PRIOR_SEED),
                   "prior_gains": np.asarray(PRIOR_GAINS, np.float64)}
            cases = []

            def make_env(finetuning, vposer_gain=1.0):
                sd = dict(vsd)
                if vposer_gain != 1.0:      # scale the embedding (last layer) so that the 11-threshold of :201 is crossed
                    for k in list(sd):
                        if k.startswith("bodyprior_enc_mu."):
                            sd[k] = sd[k] * vposer_gain
                init_env = (cfg, genop, genop, "data/smplx/models", sampler, p1, p2, pmp, fmi, markers, FakeVPoser(sd), sdf_scene_tensors(scene_free))
                return ce.CrowdEnv(init_env, save_rollout=False, render=False, finetuning=finetuning)

            def run_case(name, pair, zs, finetuning=False, obstacle_on_agent=False, goal_on_pelvis=False, steps_before=None, vposer_gain=1.0,
                         graze=False):
                env = make_env(finetuning, vposer_gain)
                sampler.sample_pairs = [(np.asarray(pair[0], np.float64), np.asarray(pair[1], np.float64))]
                torch.manual_seed(7)
                obs, _ = env.reset()
                pre = f"{name}_"
                c = {pre + "pair": np.asarray(pair, np.float32), pre + "finetuning": np.bool_(finetuning), pre + "vposer_gain": np.float64(vposer_gain),
                     pre + "z": np.stack([t2n(z) for z in zs]), pre + "reset_obs_state": t2n(obs["state"]), pre + "reset_obs_ego": t2n(obs["egosensing"]),
                     pre + "reset_obs_dist": t2n(obs["dist"]).reshape(-1), pre + "reset_obs_time": t2n(obs["time"]).reshape(-1),
                     pre + "motion_transl": t2n(env.body_scene_data["motion_seed"]["transl"]),
                     pre + "motion_glorot": t2n(env.body_scene_data["motion_seed"]["global_orient"]),
                     pre + "motion_body_pose": t2n(env.body_scene_data["motion_seed"]["body_pose"]), pre + "betas": t2n(env.betas).reshape(-1)}
                c.update(record_state(env, pre + "reset_"))
                if obstacle_on_agent:
                    # a 1 m cube on the agent's position: the same scene generator with the obstacle moved (state manipulation BEFORE
                    # the reference's step, which reads self.scene_sdf)
                    p = t2n(env.T0[0]).reshape(3)
                    lo, hi = np.array([p[0] - 0.5, p[1] - 0.5, 0.0]), np.array([p[0] + 0.5, p[1] + 0.5, 1.0])
                    env.scene_sdf = sdf_scene_tensors(synth.make_sdf_scene(RES, obstacle=(lo, hi)))
                    c[pre + "obstacle_lo"], c[pre + "obstacle_hi"] = lo.astype(np.float32), hi.astype(np.float32)
                if graze:
                    # an obstacle that only GRAZES the body: a cube centred on a marker of frame 10, its size bisected until the
                    # largest per-frame count is in (0, 40) - below the penetration threshold of :174, r_pene strictly in (0, 1)
                    import copy
                    from crowd_ppo.utils import calc_sdf
                    keep = {k: copy.deepcopy(getattr(env, k)) for k in ("state", "body_param_seed", "R0", "T0", "dist", "steps", "betas")}
                    _, loc = capture_step(ce.CrowdEnv, env, zs[0].clone())
                    for k, v in keep.items():
                        setattr(env, k, v)
                    env.flag = False
                    vw = loc["vertices_w"][0]                                        # [20, V, 3]
                    ctr = t2n(vw[10, markers[20]]).astype(np.float64)
                    lo_h, hi_h, pick = 0.02, 0.6, None
                    for _ in range(24):
                        h = 0.5 * (lo_h + hi_h)
                        sc = sdf_scene_tensors(synth.make_sdf_scene(RES, obstacle=(ctr - h, ctr + h)))
                        sv = calc_sdf(vw, sc)
                        sv[:, env.feet_vids] = 0.0
                        mx = int(sv.lt(0.0).sum(-1).max())
                        if 5 <= mx < 40:
                            pick = h
                            break
                        lo_h, hi_h = (h, hi_h) if mx < 5 else (lo_h, h)
                    assert pick is not None, "no grazing obstacle found"
                    env.scene_sdf = sdf_scene_tensors(synth.make_sdf_scene(RES, obstacle=(ctr - pick, ctr + pick)))
                    c[pre + "obstacle_lo"], c[pre + "obstacle_hi"] = (ctr - pick).astype(np.float32), (ctr + pick).astype(np.float32)
                if steps_before is not None:
                    env.steps = int(steps_before)
                    c[pre + "steps_before"] = np.int64(steps_before)
                if goal_on_pelvis:
                    # probe step on a scratch copy of the env state to learn where the pelvis ends, then put the goal there
                    import copy
                    keep = {k: copy.deepcopy(getattr(env, k)) for k in ("state", "body_param_seed", "R0", "T0", "dist", "steps", "betas")}
                    wp_keep = env.body_scene_data["wpath"].clone()
                    _, loc = capture_step(ce.CrowdEnv, env, zs[0].clone())
                    pel_w = (torch.einsum("ij,j->i", keep["R0"][0], loc["pred_pelvis_loc"][0, -1]) + keep["T0"][0, 0])
                    for k, v in keep.items():
                        setattr(env, k, v)
                    env.flag = False
                    wp = wp_keep.clone()
                    wp[1] = pel_w + torch.tensor([0.03, -0.02, 0.04])
                    env.body_scene_data["wpath"] = wp
                    c[pre + "wpath_override"] = t2n(wp)
                for i, z in enumerate(zs):
                    ret, loc = capture_step(ce.CrowdEnv, env, z.clone())
                    c.update(record_step(ret, loc, env, f"{pre}s{i}_"))
                    if ret[2]:
                        break
                c[pre + "n_steps"] = np.int64(i + 1)
                cases.append(name)
                out.update(c)
                return c

            g = torch.Generator().manual_seed(123)
            z = lambda: torch.randn(128, generator=g)   # noqa: E731
            free = [[-2.0, -1.5, 0.0], [-0.5, 2.0, 0.0]]
            run_case("free", free, [z(), z(), z()])
            run_case("pene_ft", [[-1.0, 2.0, 0.0], [2.5, 2.5, 0.0]], [z()], finetuning=True, obstacle_on_agent=True)
            run_case("pene", [[-1.0, 2.0, 0.0], [2.5, 2.5, 0.0]], [z(), z()], finetuning=False, obstacle_on_agent=True)
            run_case("graze_ft", free, [z()], finetuning=True, graze=True)
            run_case("goal", [[2.0, -2.5, 0.0], [-1.0, -2.0, 0.0]], [z()], goal_on_pelvis=True)
            run_case("depth", free, [z()], steps_before=cfg.trainconfig.max_depth - 1)
            run_case("vposer", free, [z()], vposer_gain=40.0)
            out["n_cases"] = np.int64(len(cases))
            out["cases"] = np.array(cases)
            out["cfg_json"] = np.array(json.dumps({k: cfg[k] for k in ("modelconfig", "lossconfig", "trainconfig")}))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "env_step_ref.npz"), **out)
    for name in cases:
        n = int(out[f"{name}_n_steps"])
        last = f"{name}_s{n - 1}_"
        print(f"{name:8s} steps {n} reward {float(out[last + 'reward']):+.4f} term {bool(out[last + 'terminated'])} "
              f"pene {bool(out[last + 'penetration'])} max_inside {int(out[last + 'num_inside_max'])} r_goal {float(out[last + 'r_goal'])} "
              f"r_vp {float(out[last + 'r_vp'])} vp_norm {float(out[last + 'vp_norm']):.2f}")
    print("env_step_ref", os.path.getsize(os.path.join(OUT, "env_step_ref.npz")), "bytes")


# ------------------------------------------------------------------------------------------------------------------
def _nav(scene):
    """trimesh.load(navmesh_path) of the box sampler / scene['navmesh'] of the env: vertices [n,3] float64, faces."""
    tris = np.asarray(scene["tris"], np.float32).reshape(-1, 3, 2)
    verts = np.concatenate([tris.reshape(-1, 2), np.full((tris.shape[0] * 3, 1), 1.0, np.float32)], axis=1).astype(np.float64)
    return types.SimpleNamespace(vertices=verts, faces=np.arange(tris.shape[0] * 3).reshape(-1, 3), visual=types.SimpleNamespace())


def gen_box():
    from egogen_amd import synth
    from tests.helpers import seeded_vposer_state_dict
    rec = {}
    install(rec)
    cwd = os.getcwd()
    os.chdir(REF)
    reference_first()
    try:
        with cpu_world(), tempfile.TemporaryDirectory() as td:
            from pathlib import Path
            from crowd_ppo import crowd_env_2f_box as cb
            from exp_GAMMAPrimitive.utils import environments as envs
            from models import baseops, models_GAMMA_primitive as mgp
            cfg = AttrDict(load_yaml("MPVAEPolicy_samp_collision_2.yaml"))
            bm, (p1, p2, pmp) = body_and_parsers(baseops, sys.modules["smplx"])
            fmi, markers = feet_marker_idx_and_markers()
            genop = build_combo(mgp)
            vsd = {k: v.float() for k, v in seeded_vposer_state_dict().items()}
            # two scenes of the make_box_scenes form with their obstacles in opposite corners: the middle of the floor is free, so an
            # agent that starts there stays on the 8 x 8 m floor for a few primitives
            scenes = {"s0": synth.box_scene_from_hole([-3.2, -3.2], [-2.2, -2.2]), "s1": synth.box_scene_from_hole([2.0, 2.2], [3.1, 3.0])}
            stype = "random_box_obstacle_new"
            os.makedirs(os.path.join(td, stype))
            navs = {}

            def add_scene(name, sc):
                scenes[name] = sc
                navs[name + "_navmesh_tight.ply"] = _nav(sc)
                with open(os.path.join(td, stype, name + "_shapely.pkl"), "wb") as f:
                    pickle.dump(np.asarray(sc["edges"], np.float64), f)
                with open(os.path.join(td, stype, name + "_samples.pkl"), "wb") as f:
                    pickle.dump([(np.zeros(3), np.ones(3))], f)
            for k in list(scenes):
                add_scene(k, scenes[k])
            sys.modules["trimesh"].load = lambda path, force=None: navs[os.path.basename(str(path))]
            inner = object.__new__(envs.BatchGeneratorScene2frameTrainBox)
            inner.scene_dir, inner.scene_type, inner.index_rec = Path(td), stype, 0
            inner.motion_seed_list = [os.path.join(REF, "data", "locomotion", "subseq_00343.npz")]
            inner.bm_2frame = FakeSMPLX(bm, 2)
            motion = np.load(inner.motion_seed_list[0])

            class QueueSampler:
                """The env's reset loop draws until a start passes (crowd_env_2f_box.py:349-416); the draws of a case are queued
                here and handed to the reference sampler through its own `scene_idx` / `start_target` arguments."""

                def __init__(self):
                    self.queue, self.log = [], []

                def next_body(self, **kw):
                    name, pair = self.queue.pop(0)
                    inner.scene_list = [name]
                    rec["euler_z"] = []
                    d = inner.next_body(scene_idx=0, start_target=np.asarray(pair, np.float64), **kw)
                    bp = t2n(d["motion_seed"]["body_pose"])
                    start = [s for s in range(len(motion["poses"]) - 1) if np.allclose(motion["poses"][s:s + 2, 3:66], bp, atol=1e-6)]
                    assert len(start) == 1
                    self.log.append({"scene": name, "pair": np.asarray(pair, np.float32), "start_frame": start[0], "yaw_jitter": rec["euler_z"][2],
                                     "transl": t2n(d["motion_seed"]["transl"]), "glorot": t2n(d["motion_seed"]["global_orient"]),
                                     "wpath": t2n(d["wpath"])})
                    return d
            qs = QueueSampler()
            cb.CrowdEnv._calc_egosensing = ego_hook
            out = {"body_model_seed": np.int64(0), "prior_seed": np.int64(PRIOR_SEED), "prior_gains": np.asarray(PRIOR_GAINS, np.float64),
                   }
            cases = []

            def restore(env, keep, nav, poly):
                for k, v in keep.items():
                    setattr(env, k, v)
                env.flag = False
                env.body_scene_data["navmesh"], env.scene_poly = nav, poly

            def run_case(name, draws, zs, hole=None):
                """draws: [(scene, pair)], the last one is the start the loop accepts.  hole: None | 'cover' | 'graze' - the scene is
                swapped (navmesh + polygon, like an agent moved into another scene) before the step."""
                import copy
                init_env = (cfg, genop, genop, "data/smplx/models", qs, p1, p2, pmp, fmi, markers, FakeVPoser(vsd))
                env = cb.CrowdEnv(init_env, save_rollout=False, render=False)
                qs.queue, qs.log = list(draws), []
                torch.manual_seed(11 + len(cases))
                obs, _ = env.reset()
                assert not qs.queue, "the reference's loop accepted an earlier draw than planned"
                pre = f"{name}_"
                c = {pre + "n_draws": np.int64(len(qs.log)), pre + "z": np.stack([t2n(z) for z in zs]), pre + "betas": t2n(env.betas).reshape(-1),
                     pre + "reset_obs_state": t2n(obs["state"]), pre + "reset_obs_ego": t2n(obs["egosensing"]),
                     pre + "reset_obs_dist": t2n(obs["dist"]).reshape(-1), pre + "reset_obs_time": t2n(obs["time"]).reshape(-1)}
                for i, d in enumerate(qs.log):
                    c[f"{pre}draw{i}_scene"] = np.array(d["scene"])
                    for k in ("pair", "transl", "glorot", "wpath"):
                        c[f"{pre}draw{i}_{k}"] = d[k]
                    c[f"{pre}draw{i}_start_frame"] = np.int64(d["start_frame"])
                    c[f"{pre}draw{i}_yaw_jitter"] = np.float64(d["yaw_jitter"])
                c.update(record_state(env, pre + "reset_"))
                if hole is not None:
                    keep = {k: copy.deepcopy(getattr(env, k)) for k in ("state", "body_param_seed", "R0", "T0", "dist", "steps", "betas")}
                    nav0, poly0 = env.body_scene_data["navmesh"], env.scene_poly
                    _, loc = capture_step(cb.CrowdEnv, env, zs[0].clone())
                    ctr = t2n(env.T0[0]).reshape(3)[:2].astype(np.float64)     # origin of the frame the map is sampled in
                    restore(env, keep, nav0, poly0)
                    if hole == "cover":
                        lo, hi = ctr - 0.75, ctr + 0.75
                    else:      # slide a 1 m hole towards the body until 0 < num_pene <= pene_thres (no penetration, but counted)
                        lo = hi = None
                        for off in np.linspace(1.0, 0.0, 101):     # diagonally: a corner of the hole enters the marker box first
                            l_, h_ = ctr + np.array([off, off]), ctr + np.array([off + 1.0, off + 1.0])
                            sc = synth.box_scene_from_hole(l_, h_)
                            env.body_scene_data["navmesh"], env.scene_poly = _nav(sc), np.asarray(sc["edges"], np.float64)
                            _, loc = capture_step(cb.CrowdEnv, env, zs[0].clone())
                            restore(env, keep, nav0, poly0)
                            n = float(loc["num_pene"][0])
                            if os.environ.get("EGX_GEN_DEBUG"):
                                print("graze search", round(float(off), 3), n, ctr, t2n(env.T0[0]).ravel(), float(loc["local_map"].sum()),
                                      t2n(loc["box_min"]).ravel(), t2n(loc["box_max"]).ravel(), flush=True)
                            if 0 < n <= cfg.trainconfig.pene_thres:
                                lo, hi = l_, h_
                                break
                            assert n == 0, "stepped over the (0, pene_thres] window: refine the offsets"
                        assert lo is not None, "no grazing hole found"
                    sc = synth.box_scene_from_hole(lo, hi)
                    env.body_scene_data["navmesh"], env.scene_poly = _nav(sc), np.asarray(sc["edges"], np.float64)
                    c[pre + "hole_lo"], c[pre + "hole_hi"] = lo.astype(np.float32), hi.astype(np.float32)
                for i, z in enumerate(zs):
                    ret, loc = capture_step(cb.CrowdEnv, env, z.clone())
                    c.update(record_step(ret, loc, env, f"{pre}s{i}_", box=True))
                    if ret[2]:
                        break
                c[pre + "n_steps"] = np.int64(i + 1)
                cases.append(name)
                out.update(c)

            g = torch.Generator().manual_seed(321)
            z = lambda: torch.randn(128, generator=g)   # noqa: E731
            free0 = np.array([[0.3, -0.4, 0.0], [2.5, -1.0, 0.0]], np.float32)
            free1 = np.array([[-0.5, 0.6, 0.0], [-2.5, 1.5, 0.0]], np.float32)
            on_box = np.array(free0, np.float32).copy()
            on_box[0, :2] = 0.5 * (scenes["s0"]["box_lo"] + scenes["s0"]["box_hi"])
            run_case("free", [("s0", free0)], [z(), z()])
            run_case("reject", [("s0", on_box), ("s0", on_box), ("s1", free1)], [z()])
            run_case("cover", [("s1", free1)], [z(), z()], hole="cover")
            run_case("graze", [("s0", free0)], [z()], hole="graze")
            out["cases"] = np.array(cases)
            for k in ("s0", "s1"):
                for f_ in ("tris", "edges", "box_lo", "box_hi"):
                    out[f"scene_{k}_{f_}"] = np.asarray(scenes[k][f_])
            out["cfg_json"] = np.array(json.dumps({k: cfg[k] for k in ("modelconfig", "lossconfig", "trainconfig")}))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "env_box_ref.npz"), **out)
    for name in cases:
        n = int(out[f"{name}_n_steps"])
        for i in range(n):
            k = f"{name}_s{i}_"
            print(f"{name:8s} draws {int(out[name + '_n_draws'])} step {i} reward {float(out[k + 'reward']):+.4f} term {bool(out[k + 'terminated'])} "
                  f"num_pene {float(out[k + 'num_pene'])} r_pene {float(out[k + 'r_pene'])} r_goal {float(out[k + 'r_goal'])}")
    print("env_box_ref", os.path.getsize(os.path.join(OUT, "env_box_ref.npz")), "bytes")



# ------------------------------------------------------------------------------------------------------------------
class RectPolygon:
    """The part of shapely.geometry.Polygon that crowd_env_crowd_eval.py touches (`:797-820`, `_get_dynamic_map :742-764`), for the
    only geometry that env ever builds: an axis-aligned floor square with the other members' axis-aligned marker boxes as holes.
    `contains(point)` follows shapely's definition - interior only: a point on the shell or on a hole's ring is NOT contained."""
    is_valid = True

    def __init__(self, shell, holes=None):
        def rect(ring):
            a = np.asarray(ring, np.float64).reshape(-1, 2)
            lo, hi = a.min(0), a.max(0)
            assert all(((abs(q[0] - lo[0]) < 1e-12) or (abs(q[0] - hi[0]) < 1e-12)) and ((abs(q[1] - lo[1]) < 1e-12) or (abs(q[1] - hi[1]) < 1e-12))
                       for q in a), "not an axis-aligned rectangle"
            return lo, hi
        self.ring = [tuple(map(float, q)) for q in np.asarray(shell, np.float64).reshape(-1, 2)]
        self.lo, self.hi = rect(shell)
        self.hole_rects = [rect(h) for h in (holes or [])]
        self.exterior = types.SimpleNamespace(coords=list(self.ring))

    def contains(self, pt):
        x, y = pt.xy
        if not (self.lo[0] < x < self.hi[0] and self.lo[1] < y < self.hi[1]):
            return False
        return not any(lo[0] <= x <= hi[0] and lo[1] <= y <= hi[1] for lo, hi in self.hole_rects)

    def edges(self):
        """[E,4] segments of the shell and of every hole ring: what oracle.env.calc_egosensing casts its rays against."""
        es = []
        for lo, hi in [(self.lo, self.hi)] + self.hole_rects:
            c = [(lo[0], lo[1]), (hi[0], lo[1]), (hi[0], hi[1]), (lo[0], hi[1])]
            es += [[c[q][0], c[q][1], c[(q + 1) % 4][0], c[(q + 1) % 4][1]] for q in range(4)]
        return np.asarray(es, np.float64)


class RectMulti:
    """shapely.geometry.MultiPolygon as `union_all` returns it for several boxes (`:798-802` only walks `.geoms[*].exterior`).  The
    boxes are handed on un-merged: `Polygon(floor, holes).contains` subtracts each of them, which equals subtracting their union."""

    def __init__(self, geoms):
        self.geoms = list(geoms)


class RectPoint:
    def __init__(self, xy):
        self.xy = (float(xy[0]), float(xy[1]))


def install_crowd_shapely():
    sh = sys.modules["shapely"]
    geo = sys.modules["shapely.geometry"]
    geo.Polygon, geo.Point, geo.MultiPolygon = RectPolygon, RectPoint, RectMulti
    geo.MultiPoint = geo.LinearRing = geo.mapping = object
    geo.multipolygon = types.SimpleNamespace(MultiPolygon=RectMulti)
    geo.polygon = types.SimpleNamespace(Polygon=RectPolygon)
    sh.geometry = geo
    sh.union_all = lambda polys: polys[0] if len(polys) == 1 else RectMulti(polys)
    sh.is_valid = lambda g: True
    sh.LineString = object
    pl = types.ModuleType("shapely.plotting")
    pl.plot_polygon = lambda *a, **k: None
    sys.modules["shapely.plotting"] = pl
    sh.plotting = pl


def install_vector_env_stub():
    """tianshou.env.DummyVectorEnv as crowd_ppo/dummy_vector_env.py::DummyCrowdVectorEnv uses it (tianshou 0.5 venvs.py / worker
    [upstream]): one DummyEnvWorker per env whose `send(action)` runs `env.step` at once and whose `recv()` hands the result back -
    which is what makes the reference's `for i, j in enumerate(id): update_holes...; workers[j].send(action[i])` loop sequential."""
    class Worker:
        def __init__(self, fn):
            self.env, self.result = fn(), None

        def send(self, action):
            self.result = self.env.reset() if action is None else self.env.step(action)

        def recv(self):
            return list(self.result)

    class DummyVectorEnv:
        def __init__(self, env_fns, **kw):
            self._env_fns = env_fns
            self.workers = [Worker(fn) for fn in env_fns]
            self.env_num, self.is_async, self.is_closed = len(env_fns), False, False

        def _assert_is_not_closed(self):
            assert not self.is_closed

        def _wrap_id(self, id=None):
            return list(range(self.env_num)) if id is None else ([id] if np.isscalar(id) else list(id))

        def get_env_attr(self, key, id=None):
            return [getattr(self.workers[j].env, key) for j in self._wrap_id(id)]

        def set_env_attr(self, key, value, id=None):
            for j in self._wrap_id(id):
                setattr(self.workers[j].env, key, value)

        def reset(self, id=None):
            return [self.workers[j].env.reset() for j in self._wrap_id(id)]
    te = types.ModuleType("tianshou.env")
    te.DummyVectorEnv = DummyVectorEnv
    tu = types.ModuleType("tianshou.env.utils")
    tu.ENV_TYPE, tu.gym_new_venv_step_type = object, tuple
    tw = types.ModuleType("tianshou.env.worker")
    tw.DummyEnvWorker = tw.EnvWorker = tw.RayEnvWorker = tw.SubprocEnvWorker = Worker
    ts = sys.modules.get("tianshou") or types.ModuleType("tianshou")
    ts.env = te
    sys.modules.update({"tianshou": ts, "tianshou.env": te, "tianshou.env.utils": tu, "tianshou.env.worker": tw})


def crowd_ego_hook(self, joint):
    """Stands in for crowd_env_crowd_eval.CrowdEnv._calc_egosensing (shapely ray casting, identical to crowd_env_2f's): the rays are
    cast against the rings of the polygon `_get_feature` has just built (floor + the holes this member sees NOW); snapshots the
    caller's locals like ego_hook, plus the holes."""
    from oracle.env import calc_egosensing
    fr = sys._getframe(1)
    if fr.f_code.co_name == "step":
        _STEP_LOCALS.clear()
        _STEP_LOCALS.update(fr.f_locals)
        _STEP_LOCALS["holes_seen"] = np.asarray(self.holes, np.float64).copy()
    return calc_egosensing(joint, self.scene_poly.edges()).float()


def gen_crowd():
    """BASELINE config 5's plumbing, executed: four `crowd_env_crowd_eval.CrowdEnv` members (constructor box `:54-75`, `reset
    :384-454`, `step :102-382` with `_get_feature :766-837` / `_get_dynamic_map :742-764`) built from `CrowdMotion.next_body`
    (`environments.py:1041-1157`) and driven through the reference's own `DummyCrowdVectorEnv` (`dummy_vector_env.py:29-128`: the
    holes of every member are refreshed before EACH member's step).  -> tests/golden/env_crowd_ref.npz"""
    from egogen_amd import synth
    from tests.helpers import seeded_vposer_state_dict
    rec = {}
    install(rec)
    install_crowd_shapely()
    install_vector_env_stub()
    cwd = os.getcwd()
    os.chdir(REF)
    reference_first()
    try:
        with cpu_world():
            from crowd_ppo import crowd_env_crowd_eval as cc
            from crowd_ppo.dummy_vector_env import DummyCrowdVectorEnv
            from exp_GAMMAPrimitive.utils import environments as envs
            from models import baseops, models_GAMMA_primitive as mgp
            cfg = AttrDict(load_yaml("MPVAEPolicy_samp_collision_2.yaml"))      # main_crowd_eval.py:224 load_model(box=True)
            cfg["args"] = {"gpu_index": 0}
            bm, (p1, p2, pmp) = body_and_parsers(baseops, sys.modules["smplx"])
            fmi, markers = feet_marker_idx_and_markers()
            genop = build_combo(mgp)
            vsd = {k: v.float() for k, v in seeded_vposer_state_dict().items()}
            sampler = object.__new__(envs.CrowdMotion)
            sampler.bm_male = sampler.bm_female = FakeSMPLX(bm, 2)
            motion = np.load(os.path.join(REF, "data", "locomotion", "subseq_00343.npz"))
            cc.CrowdEnv._calc_egosensing = crowd_ego_hook
            G = 4
            out = {"body_model_seed": np.int64(0), "prior_seed": np.int64(PRIOR_SEED), "prior_gains": np.asarray(PRIOR_GAINS, np.float64),
                   "G": np.int64(G)}
            cases = []

            def run_case(name, radius, phase, n_rounds, seed):
                # main_crowd_eval.py:276-283: four points on a circle, every member walks to the opposite one
                t = np.linspace(phase, phase + 2 * np.pi, G, endpoint=False)
                pts = np.zeros((G, 3))
                pts[:, 0], pts[:, 1] = radius * np.cos(t), radius * np.sin(t)
                start_target = [(pts[k], pts[(k + 2) % G]) for k in range(G)]
                torch.manual_seed(seed)
                rec["euler_z"] = []
                datas = sampler.next_body(start_target=start_target, fixed_seed=True, num_agents=G)
                pre = f"{name}_"
                c = {pre + "start_target": np.asarray(start_target, np.float32)}
                for k, d in enumerate(datas):
                    bp = t2n(d["motion_seed"]["body_pose"])
                    start = [s_ for s_ in range(len(motion["poses"]) - 1) if np.allclose(motion["poses"][s_:s_ + 2, 3:66], bp, atol=1e-6)]
                    assert len(start) == 1
                    c[f"{pre}m{k}_start_frame"] = np.int64(start[0])
                    c[f"{pre}m{k}_yaw_jitter"] = np.float64(rec["euler_z"][k])
                    c[f"{pre}m{k}_transl"], c[f"{pre}m{k}_glorot"] = t2n(d["motion_seed"]["transl"]), t2n(d["motion_seed"]["global_orient"])
                    c[f"{pre}m{k}_wpath"] = t2n(d["wpath"])
                assert len(rec["euler_z"]) == G
                members = [cc.CrowdEnv([cfg, genop, genop, "data/smplx/models", datas[k], p1, p2, pmp, fmi, markers, FakeVPoser(vsd), str(k), name],
                                       save_rollout=False, render=False) for k in range(G)]
                for k, m in enumerate(members):
                    c[f"{pre}m{k}_init_bbox"] = np.asarray(m.bbox, np.float64)
                venv = DummyCrowdVectorEnv([(lambda m=m: m) for m in members])
                for k, m in enumerate(members):
                    c[f"{pre}m{k}_init_holes"] = np.asarray(m.holes, np.float64)
                obs = venv.reset()
                for k, (o, _) in enumerate(obs):
                    m = members[k]
                    c.update({f"{pre}m{k}_reset_obs_state": t2n(o["state"]), f"{pre}m{k}_reset_obs_ego": t2n(o["egosensing"]),
                              f"{pre}m{k}_reset_obs_dist": t2n(o["dist"]).reshape(-1), f"{pre}m{k}_reset_obs_time": t2n(o["time"]).reshape(-1),
                              f"{pre}m{k}_betas": t2n(m.betas).reshape(-1)})
                    c.update(record_state(m, f"{pre}m{k}_reset_"))
                g = torch.Generator().manual_seed(seed + 1000)
                zs = torch.randn(n_rounds, G, 128, generator=g)
                c[pre + "z"] = t2n(zs)
                # the reference's vector step, with the members' step wrapped so that each one's locals are kept
                caught = {}
                for k, m in enumerate(members):
                    def wrapped(action, m=m, k=k, orig=m.step):
                        _STEP_LOCALS.clear()
                        ret = orig(action)
                        caught[k] = (ret, dict(_STEP_LOCALS))
                        return ret
                    m.step = wrapped
                alive = True
                for r in range(n_rounds):
                    caught.clear()
                    o_, rew_, term_, trunc_, info_ = venv.step(t2n(zs[r]))
                    for k in range(G):
                        ret, loc = caught[k]
                        sp = f"{pre}r{r}_m{k}_"
                        c.update(record_step(ret, loc, members[k], sp, box=True))
                        c[sp + "holes_seen"] = loc["holes_seen"]
                        c[sp + "bbox_after"] = np.asarray(members[k].bbox, np.float64)
                        c[sp + "local_map"] = t2n(loc["local_map"][0])
                        assert float(rew_[k]) == float(ret[1]) and bool(term_[k]) == bool(ret[2])
                    if term_.any():
                        alive = False
                        c[pre + "n_rounds"] = np.int64(r + 1)
                        break
                if alive:
                    c[pre + "n_rounds"] = np.int64(n_rounds)
                cases.append(name)
                out.update(c)

            run_case("ring", 1.2, 0.3, 2, 31)       # close: rays hit the others' boxes, marker boxes overlap walk-map cells late
            run_case("tight", 0.55, 1.1, 2, 57)     # shoulder to shoulder: the others' boxes cover cells of the local map
            out["cases"] = np.array(cases)
            out["cfg_json"] = np.array(json.dumps({k: cfg[k] for k in ("modelconfig", "lossconfig", "trainconfig")}))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "env_crowd_ref.npz"), **out)
    for name in cases:
        for r in range(int(out[name + "_n_rounds"])):
            for k in range(G):
                sp = f"{name}_r{r}_m{k}_"
                print(f"{name:6s} round {r} member {k} reward {float(out[sp + 'reward']):+.4f} term {bool(out[sp + 'terminated'])} "
                      f"num_pene {float(out[sp + 'num_pene'])} r_pene {float(out[sp + 'r_pene'])} ego min {float(out[sp + 'obs_ego'].min()):+.3f} "
                      f"map -1 cells {int((out[sp + 'local_map'] < 0).sum())}")
    print("env_crowd_ref", os.path.getsize(os.path.join(OUT, "env_crowd_ref.npz")), "bytes")



# ------------------------------------------------------------------------------------------------------------------
def gen_rollout():
    """motion/vis.py::rollout_primitives (:44-78; the same function in vis_crowd.py:43 and experiments/gen_egobody_depth.py:27-61),
    EXECUTED on a seeded motion list of the `log/eval_results/motion_*.pkl` form (three primitives: first, '2-frame', '1-frame').
    smplx.create -> FakeSMPLX (the function only reads the rest-pose pelvis of the primitive's betas from it); trimesh / pyrender /
    tqdm are imported by the module and unused by the function.  -> tests/golden/rollout_prims_ref.npz"""
    from scipy.spatial.transform import Rotation
    rec = {}
    install(rec)
    for name in ("tqdm",):
        sys.modules.setdefault(name, types.ModuleType(name))
    cwd = os.getcwd()
    os.chdir(REF)
    reference_first()
    try:
        with cpu_world():
            from egogen_amd import synth
            from oracle.smplx_lbs import BodyModel
            bm = BodyModel(synth.make_body_model(0))
            sys.modules["smplx"].create = lambda *a, batch_size=1, **k: FakeSMPLX(bm, batch_size)
            # vis.py is a script (it parses arguments and opens a viewer at import): its head - the imports and the function, up
            # to the next top-level `def` - is executed as it stands; `model_path` is the global its tail would have set (:417)
            src = open(os.path.join(REF, "vis.py")).read()
            head = src[:src.index("\ndef vis_results_new(")]
            assert "def rollout_primitives(motion_primitives):" in head
            ns = {"__name__": "vis_head", "model_path": "data/smplx/models"}
            exec(compile(head, os.path.join(REF, "vis.py"), "exec"), ns)
            vis = types.SimpleNamespace(rollout_primitives=ns["rollout_primitives"])
            g = torch.Generator().manual_seed(808)
            betas = torch.randn(10, generator=g).numpy()
            mps = []
            for i, kind in enumerate(("2-frame", "2-frame", "1-frame")):
                xb = torch.zeros(20, 93)
                xb[:, :3] = torch.randn(20, 3, generator=g) * 0.5
                xb[:, 3:6] = torch.randn(20, 3, generator=g) * 0.7
                xb[:, 6:69] = torch.randn(20, 63, generator=g) * 0.2
                xb[:, 69:] = torch.randn(20, 24, generator=g) * 0.3
                R = Rotation.from_euler("zyx", (torch.rand(3, generator=g).numpy() - 0.5) * [6.0, 0.4, 0.4]).as_matrix().astype(np.float32)
                mps.append({"smplx_params": xb[None].numpy(), "betas": betas, "gender": "male", "transf_rotmat": R,
                            "transf_transl": (torch.randn(1, 3, generator=g) * 2).numpy(), "mp_type": kind})
            out = {"n": np.int64(len(mps)), "betas": betas, "body_model_seed": np.int64(0)}
            for i, mp in enumerate(mps):
                out[f"p{i}_smplx_params"], out[f"p{i}_rotmat"], out[f"p{i}_transl"] = mp["smplx_params"].copy(), mp["transf_rotmat"], mp["transf_transl"]
                out[f"p{i}_mp_type"] = np.array(mp["mp_type"])
            import copy
            seq = vis.rollout_primitives(copy.deepcopy(mps))       # (it rewrites the primitives' parameter rows in place)
            out["sequence"] = np.asarray(seq, np.float64)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "rollout_prims_ref.npz"), **out)
    print("rollout_prims_ref", out["sequence"].shape, os.path.getsize(os.path.join(OUT, "rollout_prims_ref.npz")), "bytes")



# ------------------------------------------------------------------------------------------------------------------
class RingPolygon:
    """shapely.geometry.Polygon as crowd_env_egobody_eval.py / environments.py::Egobody touch it, for general rings (the walkable
    region of a navmesh: one exterior, several interior rings).  `contains` = interior (even-odd over all rings, oracle.env.
    points_in_rings).  Constructing a Polygon FROM a Polygon returns that polygon (shapely 2.0 geometry/polygon.py, `Polygon.__new__`:
    "return original objects since geometries are immutable") - which is why `Polygon(self.scene_poly, holes)` (:824) leaves the
    other person's box out of the scene: an assumption about shapely, NOT something this fixture can pin."""
    is_valid = True

    def __new__(cls, shell=None, holes=None):
        if isinstance(shell, RingPolygon):
            return shell
        return object.__new__(cls)

    def __init__(self, shell=None, holes=None):
        if isinstance(shell, RingPolygon):
            return
        def closed(r):
            a = np.asarray(r, np.float64).reshape(-1, np.asarray(r).shape[-1])[:, :2]
            return a if np.array_equal(a[0], a[-1]) else np.concatenate([a, a[:1]], 0)
        self.rings = [closed(shell)] + [closed(h) for h in (holes or [])]
        self.exterior = types.SimpleNamespace(coords=[tuple(map(float, q)) for q in self.rings[0]])

    def edges(self):
        return np.concatenate([np.concatenate([r[:-1], r[1:]], 1) for r in self.rings], 0)

    def contains(self, pt):
        from oracle.env import points_in_rings
        x, y = pt.xy
        return bool(points_in_rings(self.edges(), np.array([x], np.float64), np.array([y], np.float64))[0])


def gen_egobody():
    """The EgoBody evaluation, executed: two `crowd_env_egobody_eval.CrowdEnv` members (`reset :395-466`, `step
    :102-393` with the pelvis filter `:208-216`, the pose filter at 14 `:229-234`, termination at max_depth only `:378`, `_get_feature
    :777-850` on the scene's walkable polygon) built from `Egobody.gen_init_body` (`environments.py:679-765`: random start frame, random
    betas ~ N(0, 0.3^2), +-0.2 x 2 pi yaw - all recorded) inside the walkable region of the in-tree Replica room_0 navmesh, under the
    reference's `DummyCrowdVectorEnv`.  -> tests/golden/env_egobody_ref.npz"""
    from egogen_amd import synth
    from egogen_amd.egobody import EgobodySampler
    from tests.helpers import seeded_vposer_state_dict
    rec = {}
    install(rec)
    install_crowd_shapely()
    geo = sys.modules["shapely.geometry"]
    geo.Polygon = RingPolygon
    geo.polygon = types.SimpleNamespace(Polygon=RingPolygon)
    install_vector_env_stub()
    cwd = os.getcwd()
    os.chdir(REF)
    reference_first()
    try:
        with cpu_world():
            from crowd_ppo import crowd_env_egobody_eval as ce
            from crowd_ppo.dummy_vector_env import DummyCrowdVectorEnv
            from exp_GAMMAPrimitive.utils import environments as envs
            from models import baseops, models_GAMMA_primitive as mgp
            cfg = AttrDict(load_yaml("MPVAEPolicy_samp_collision_2.yaml"))      # main_egobody_eval.py:215 load_model(box=True)
            cfg["args"] = {"gpu_index": 0}
            bm, (p1, p2, pmp) = body_and_parsers(baseops, sys.modules["smplx"])
            fmi, markers = feet_marker_idx_and_markers()
            genop = build_combo(mgp)
            vsd = {k: v.float() for k, v in seeded_vposer_state_dict().items()}
            a = synth.load_assets()
            prod = EgobodySampler(a["room0_nav_v"], a["room0_nav_f"], [{"poses": a["seed_poses"], "trans": a["seed_trans"]}], seed=3)
            rings = [np.asarray(r, np.float64) for r in prod.rings]
            areas = [abs(0.5 * float(np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))) for r in rings]
            assert int(np.argmax(areas)) == 0, "ring 0 is the exterior"
            sampler = object.__new__(envs.Egobody)
            sampler.bm_male = sampler.bm_female = FakeSMPLX(bm, 2)
            sampler.motion_seed_list = [os.path.join(REF, "data", "locomotion", "subseq_00343.npz")]
            sampler.scene_dir = "data/room_0"
            sampler.navmesh = types.SimpleNamespace(vertices=np.asarray(a["room0_nav_v"]), faces=np.asarray(a["room0_nav_f"]))
            sampler.walkable_region = RingPolygon(rings[0], rings[1:])
            motion = np.load(sampler.motion_seed_list[0])
            ce.CrowdEnv._calc_egosensing = lambda self, joint: _ego_dyn(self, joint)
            out = {"body_model_seed": np.int64(0), "prior_seed": np.int64(PRIOR_SEED), "prior_gains": np.asarray(PRIOR_GAINS, np.float64),
                   "G": np.int64(2), "n_rings": np.int64(len(rings))}
            for i, r in enumerate(rings):
                out[f"ring{i}"] = r
            cases = []

            def _ego_dyn(self, joint):
                from oracle.env import calc_egosensing
                fr = sys._getframe(2)
                if fr.f_code.co_name == "step":
                    _STEP_LOCALS.clear()
                    _STEP_LOCALS.update(fr.f_locals)
                    _STEP_LOCALS["holes_seen"] = np.asarray(self.holes, np.float64).copy()
                return calc_egosensing(joint, self.scene_poly_dyn.edges()).float()

            def run_case(name, pair, n_rounds, seed, vposer_gain=1.0, z_scale=1.0, expect_exit=None):
                """pair: (start, target) of member 0; member 1 walks the opposite way (Egobody.next_body :780-782)."""
                start, target = np.asarray(pair[0], np.float64), np.asarray(pair[1], np.float64)
                torch.manual_seed(seed)
                random.seed(seed)
                rec["euler_z"] = []
                datas = [sampler.gen_init_body(start, target, "male"), sampler.gen_init_body(target, start, "male")]
                pre = f"{name}_"
                c = {pre + "start_target": np.asarray([[start, target], [target, start]], np.float32), pre + "vposer_gain": np.float64(vposer_gain)}
                for k, d in enumerate(datas):
                    bp = t2n(d["motion_seed"]["body_pose"])
                    st_ = [s_ for s_ in range(len(motion["poses"]) - 1) if np.allclose(motion["poses"][s_:s_ + 2, 3:66], bp, atol=1e-6)]
                    assert len(st_) == 1
                    c[f"{pre}m{k}_start_frame"] = np.int64(st_[0])
                    c[f"{pre}m{k}_yaw_jitter"] = np.float64(rec["euler_z"][k])
                    c[f"{pre}m{k}_betas"] = t2n(d["betas"]).reshape(-1)
                    c[f"{pre}m{k}_transl"], c[f"{pre}m{k}_glorot"] = t2n(d["motion_seed"]["transl"]), t2n(d["motion_seed"]["global_orient"])
                    c[f"{pre}m{k}_wpath"] = t2n(d["wpath"])
                sd = dict(vsd)
                if vposer_gain != 1.0:
                    for kk in list(sd):
                        if kk.startswith("bodyprior_enc_mu."):
                            sd[kk] = sd[kk] * vposer_gain
                members = [ce.CrowdEnv([cfg, genop, genop, "data/smplx/models", datas[k], p1, p2, pmp, fmi, markers, FakeVPoser(sd), str(k), "tmp"],
                                       save_rollout=False, render=False) for k in range(2)]
                for k, m in enumerate(members):
                    c[f"{pre}m{k}_init_bbox"] = np.asarray(m.bbox, np.float64)
                venv = DummyCrowdVectorEnv([(lambda m=m: m) for m in members])
                obs = venv.reset()
                for k, (o, _) in enumerate(obs):
                    m = members[k]
                    c.update({f"{pre}m{k}_reset_obs_state": t2n(o["state"]), f"{pre}m{k}_reset_obs_ego": t2n(o["egosensing"]),
                              f"{pre}m{k}_reset_obs_dist": t2n(o["dist"]).reshape(-1), f"{pre}m{k}_reset_obs_time": t2n(o["time"]).reshape(-1)})
                    c.update(record_state(m, f"{pre}m{k}_reset_"))
                g = torch.Generator().manual_seed(seed + 1000)
                zs = torch.randn(n_rounds, 2, 128, generator=g) * z_scale
                c[pre + "z"] = t2n(zs)
                caught = {}
                for k, m in enumerate(members):
                    def wrapped(action, m=m, k=k, orig=m.step):
                        _STEP_LOCALS.clear()
                        ret = orig(action)
                        caught[k] = (ret, dict(_STEP_LOCALS))
                        return ret
                    m.step = wrapped
                exit_at = None
                import contextlib, io
                for r in range(n_rounds):
                    caught.clear()
                    buf = io.StringIO()
                    try:
                        with contextlib.redirect_stdout(buf):
                            venv.step(t2n(zs[r]))
                    except SystemExit:
                        k_exit = len(caught)                         # members before it finished their step of this round
                        reason = buf.getvalue().strip().splitlines()[-1]
                        exit_at = (r, k_exit, reason)
                    for k in sorted(caught):
                        ret, loc = caught[k]
                        sp = f"{pre}r{r}_m{k}_"
                        c.update(record_step(ret, loc, members[k], sp, box=True))
                        c[sp + "holes_seen"] = loc["holes_seen"]
                        c[sp + "bbox_after"] = np.asarray(members[k].bbox, np.float64)
                        c[sp + "local_map"] = t2n(loc["local_map"][0])
                    if exit_at is not None:
                        break
                c[pre + "n_rounds"] = np.int64(r + 1 if exit_at is None else exit_at[0] + (1 if exit_at[1] > 0 else 0))
                c[pre + "exit_round"], c[pre + "exit_member"] = np.int64(-1 if exit_at is None else exit_at[0]), np.int64(-1 if exit_at is None else exit_at[1])
                c[pre + "exit_reason"] = np.array("" if exit_at is None else exit_at[2])
                if expect_exit is not None:
                    assert exit_at is not None and expect_exit in exit_at[2], (name, exit_at)
                else:
                    assert exit_at is None, (name, exit_at)
                cases.append(name)
                out.update(c)
                return exit_at

            # starts / targets with 0.3 m of clearance from the product's sampler (Egobody.next_body draws them with trimesh + shapely)
            pairs = [prod.next_body() for _ in range(6)]
            wp = [np.asarray(pp[0]["wpath"], np.float64) for pp in pairs]
            def attempt(name, *a_, **k_):
                """run_case, undone if its expectation about the exit fails (the random-init prior decides where the body walks)"""
                try:
                    return True, run_case(name, *a_, **k_)
                except AssertionError:
                    if cases and cases[-1] == name:
                        cases.pop()
                    for kk in [k2 for k2 in out if k2.startswith(name + "_")]:
                        del out[kk]
                    return False, None
            clear = None
            for i in range(len(wp)):                      # a pair whose first two primitives stay inside the region
                ok, _ = attempt("pair", wp[i], 2, 71 + i)
                if ok:
                    clear = i
                    break
            assert clear is not None
            ok, _ = attempt("pose", wp[clear], 1, 71 + clear, vposer_gain=40.0, expect_exit="unrealistic pose")
            assert ok
            # a walk that leaves the region (into a furniture hole or through a wall) within the first five steps
            found = None
            for trial in range(12):
                ok, e = attempt("pelvis", wp[trial % len(wp)], 5, 90 + trial, expect_exit="invalid pelvis location")
                if ok and (found is None):
                    found = e
                    if e[0] >= 1 or trial >= 6:       # prefer an exit after at least one complete round
                        break
                    if trial < 6:
                        cases.pop()
                        for kk in [k2 for k2 in out if k2.startswith("pelvis_")]:
                            del out[kk]
                        found = None
            assert found is not None, "no trajectory left the walkable region within five steps"
            out["cases"] = np.array(cases)
            out["cfg_json"] = np.array(json.dumps({k: cfg[k] for k in ("modelconfig", "lossconfig", "trainconfig")}))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "env_egobody_ref.npz"), **out)
    for name in cases:
        print(name, "rounds", int(out[name + "_n_rounds"]), "exit", int(out[name + "_exit_round"]), int(out[name + "_exit_member"]), str(out[name + "_exit_reason"]))
        for r in range(int(out[name + "_n_rounds"])):
            for k in range(2):
                sp = f"{name}_r{r}_m{k}_"
                if sp + "reward" in out:
                    print(f"   round {r} member {k} reward {float(out[sp + 'reward']):+.4f} term {bool(out[sp + 'terminated'])} num_pene {float(out[sp + 'num_pene'])} "
                          f"r_vp {float(out[sp + 'r_vp'])} ego min {float(out[sp + 'obs_ego'].min()):+.3f} map -1 cells {int((out[sp + 'local_map'] < 0).sum())}")
    print("env_egobody_ref", os.path.getsize(os.path.join(OUT, "env_egobody_ref.npz")), "bytes")


# ------------------------------------------------------------------------------------------------------------------
def install_tianshou_stub(record):
    """The attributes of tianshou 0.5 that GAMMAPPOPolicy.__init__ / learn read (policy/base.py, modelfree/pg.py, a2c.py, ppo.py,
    data/batch.py [upstream]) - the loss, the backward pass, the clip and the optimiser step executed are ppo_policy.py's own lines."""
    class Batch:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __len__(self):
            return len(self.act)

        def __getitem__(self, idx):
            def pick(v):
                if isinstance(v, dict):
                    return {k: pick(x) for k, x in v.items()}
                return v[idx]
            return Batch(**{k: pick(v) for k, v in self.__dict__.items()})

        def split(self, size, shuffle=True, merge_last=False):
            """tianshou.data.Batch.split: a random permutation, chunks of `size`, the remainder merged into the last chunk."""
            length = len(self)
            indices = np.random.permutation(length) if shuffle else np.arange(length)
            record.setdefault("perms", []).append(indices.copy())
            merge_last = merge_last and length % size > 0
            for idx in range(0, length, size):
                if merge_last and idx + size + size >= length:
                    yield self[indices[idx:]]
                    break
                yield self[indices[idx:idx + size]]

    class _AC(torch.nn.Module):                      # tianshou.utils.net.common.ActorCritic: actor + critic ONLY
        def __init__(self, actor, critic):
            super().__init__()
            self.actor, self.critic = actor, critic

    class PPOPolicy(torch.nn.Module):
        """BasePolicy -> PGPolicy -> A2CPolicy -> PPOPolicy constructor chain, reduced to the attributes it leaves behind."""

        def __init__(self, actor, critic, optim, dist_fn, discount_factor=0.99, gae_lambda=0.95, max_grad_norm=None, vf_coef=0.5,
                     ent_coef=0.01, reward_normalization=False, max_batchsize=256, deterministic_eval=False, action_space=None,
                     action_scaling=True, action_bound_method="clip", lr_scheduler=None, **kw):
            super().__init__()
            self.actor, self.critic, self.optim, self.dist_fn = actor, critic, optim, dist_fn
            self._gamma, self._lambda, self._grad_norm = discount_factor, gae_lambda, max_grad_norm
            self._weight_vf, self._weight_ent, self._rew_norm, self._batch = vf_coef, ent_coef, reward_normalization, max_batchsize
            self._deterministic_eval, self._eps = deterministic_eval, 1e-8
            self._actor_critic = _AC(self.actor, self.critic)       # a2c.py: what clip_grad_norm_ of :244-247 sees
    ts = gg._stub("tianshou")
    ts.data = gg._stub("tianshou.data", Batch=Batch, ReplayBuffer=object, to_torch_as=lambda x, y: torch.as_tensor(x).to(y))
    ts.policy = gg._stub("tianshou.policy", PPOPolicy=PPOPolicy)
    return Batch


def gen_learn():
    from torch.distributions import Independent, Normal
    rec = {}
    gg.install_stubs()
    Batch = install_tianshou_stub(rec)
    cwd = os.getcwd()
    os.chdir(REF)
    reference_first()
    try:
        from crowd_ppo.ppo_policy import GAMMAPPOPolicy
        from models.models_policy_ppo import ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase
        cfg = load_yaml("MPVAEPolicy_samp_collision.yaml")["modelconfig"]
        out = {}
        for case, kw in (("default", dict()), ("options", dict(value_clip=1, dual_clip=2.0))):
            torch.manual_seed(3)
            actor, critic, base = GAMMAActor(cfg), GAMMACritic(cfg), GAMMAPolicyBase(cfg)
            gg.fill_module(base, seed=102)
            gg.fill_module(actor, seed=103, gain=1.4)
            gg.fill_module(critic, seed=104, gain=1.4)
            for m in actor.pnet.modules():                         # main_ppo.py:128-131: last-policy-layer scaling (keeps sigma, ratio sane)
                if isinstance(m, torch.nn.Linear):
                    m.weight.data.mul_(0.05)
            ac = ActorCritic(actor, critic, base)
            key_shapes = {}
            for pre_, mod in (("shared_net.", base), ("actor.", actor), ("critic.", critic)):
                for k, v in mod.state_dict().items():
                    key_shapes[pre_ + k] = tuple(v.shape)
            optim = torch.optim.AdamW(ac.parameters(), lr=3e-4, weight_decay=0.01)
            pol = GAMMAPPOPolicy(actor, critic, base, optim, lambda *l: Independent(Normal(*l), 1), discount_factor=0.99, gae_lambda=0.95,
                                 max_grad_norm=0.1, vf_coef=1.0, ent_coef=0.01, weight_kld=0, reward_normalization=False, action_space=None,
                                 action_scaling=False, action_bound_method="", eps_clip=0.1, value_clip=kw.get("value_clip", 0),
                                 dual_clip=kw.get("dual_clip"), advantage_normalization=1, recompute_advantage=0, deterministic_eval=False)
            assert not hasattr(pol, "_actor_critic") or not any(p is q for p in pol._actor_critic.parameters() for q in base.parameters()), \
                "ppo_policy.py:88 is an annotation: the clipped set is the parent's actor + critic"
            g = torch.Generator().manual_seed(500)
            N, BS = 160, 64                       # minibatches of 64 and 96 rows (merge_last)
            obs = {"state": torch.randn(N, 2, 402, generator=g) * 0.5, "egosensing": torch.rand(N, 2, 32, generator=g) * 2 - 1,
                   "dist": torch.rand(N, generator=g), "time": torch.rand(N, generator=g)}
            with torch.no_grad():
                hx = base(obs)
                (mu, lv), _ = actor(hx)
                sig = torch.exp(lv.clamp(-2.5, 2.5)) ** 0.5
                act = mu + sig * torch.randn(N, 128, generator=g)
                dist0 = Independent(Normal(mu, sig), 1)
                logp_old = dist0.log_prob(act) + 0.05 * torch.randn(N, generator=g)     # ratios on both sides of the clip range
                v_s = critic(hx).flatten()
            adv = torch.randn(N, generator=g) * 2.0
            ret = v_s + adv + 0.3 * torch.randn(N, generator=g)
            batch = Batch(obs=obs, act=act, adv=adv.clone(), returns=ret, logp_old=logp_old, v_s=v_s, z_mu=mu, info={})
            p0 = {k: v.detach().clone() for k, v in ac.state_dict().items()}
            steps = []
            real_step = optim.step

            def spy_step(*a, **k):
                steps.append({n_: p_.grad.detach().clone() for n_, p_ in ac.named_parameters()})
                return real_step(*a, **k)
            optim.step = spy_step
            np.random.seed(77)
            rec["perms"] = []
            res = pol.learn(batch, BS, 1)
            pre = case + "_"
            names = [n_ for n_, _ in ac.named_parameters()]
            out.update({pre + "fill_seeds": np.array([102, 103, 104], np.int64), pre + "fill_gains": np.array([1.0, 1.4, 1.4]),
                        pre + "pnet_scale": np.float64(0.05), pre + "state_dict_keys": np.array(list(key_shapes.keys())),
                        pre + "state_dict_shapes": np.array([str(v) for v in key_shapes.values()]), pre + "batch_size": np.int64(BS),
                        pre + "perm": rec["perms"][0].astype(np.int64), pre + "act": act.numpy(), pre + "adv": adv.numpy(), pre + "returns": ret.numpy(),
                        pre + "logp_old": logp_old.numpy(), pre + "v_s": v_s.numpy(), pre + "z_mu": mu.numpy(),
                        pre + "value_clip": np.int64(kw.get("value_clip", 0)), pre + "dual_clip": np.float64(kw.get("dual_clip") or 0.0),
                        pre + "param_names": np.array(names)})
            out.update({pre + "obs_" + k: v.numpy() for k, v in obs.items()})
            for k, v in res.items():
                out[pre + "res_" + k.replace("/", "_")] = np.asarray(v, np.float64)
            for i, gr in enumerate(steps):           # the gradients the optimiser saw (after the actor + critic clip of :244-247)
                out[f"{pre}mb{i}_grad_norm"] = np.array([float(gr[n_].norm()) for n_ in names])
                out[f"{pre}mb{i}_grad_head"] = np.stack([np.resize(gr[n_].flatten()[:8].numpy(), 8) for n_ in names])
                ac_names = [n_ for n_ in names if not n_.startswith("shared_net.")]
                out[f"{pre}mb{i}_clipped_set_norm"] = np.float64(torch.sqrt(sum(gr[n_].pow(2).sum() for n_ in ac_names)))
            p1 = ac.state_dict()
            out[pre + "delta_norm"] = np.array([float((p1[n_] - p0[n_]).norm()) for n_ in names])
            out[pre + "delta_head"] = np.stack([np.resize((p1[n_] - p0[n_]).flatten()[:8].numpy(), 8) for n_ in names])
            out[pre + "param_head_after"] = np.stack([np.resize(p1[n_].flatten()[:8].numpy(), 8) for n_ in names])
            print(case, {k: np.round(v, 5).tolist() for k, v in res.items()}, "clipped-set norms", [float(out[f"{pre}mb{i}_clipped_set_norm"]) for i in range(len(steps))],
                  "shared_net grad norm", float(np.sqrt(sum(float(steps[0][n_].pow(2).sum()) for n_ in names if n_.startswith("shared_net.")))))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "ppo_learn_ref.npz"), **out)
    print("ppo_learn_ref", os.path.getsize(os.path.join(OUT, "ppo_learn_ref.npz")), "bytes")


TARGETS = {"sdf": gen_sdf, "box": gen_box, "learn": gen_learn, "crowd": gen_crowd, "rollout": gen_rollout, "egobody": gen_egobody}

if __name__ == "__main__":
    which = sys.argv[1:] or ["sdf"]
    if which == ["all"]:
        which = list(TARGETS)
    if len(which) == 1:
        TARGETS[which[0]]()
    else:
        # every target installs its own substitutes over sys.modules (smplx.create, shapely, the tianshou stand-in ...) and takes
        # the repository root off sys.path: one interpreter per target
        import subprocess
        for w in which:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
