"""GPU tests of the drop-in boundary beyond the step loop: the SMPLXParser methods the environment's callers
use (get_new_coordinate, calc_calibrate_offset, update_transl_glorot), the stand-alone _get_feature / get_map operators
against reference-generated goldens, and BASELINE configs[0] - the real Replica room0 walkable polygon (6 rings, concave
exterior, holes) and its start/target pairs through the single-agent CrowdEnv view."""
import ctypes as C
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from egogen_amd import synth
from tests.helpers import build_world, load_golden, max_abs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parser(V=1536, seed=0):
    from egogen_amd.body_model import BodyModelHandle, SMPLXParser
    from oracle.smplx_lbs import BodyModel
    bm = synth.make_body_model(seed, num_verts=V)
    mk, feet = synth.marker_ids(V), synth.feet_vids(V)
    h = BodyModelHandle(bm, mk, feet)
    p = SMPLXParser({"n_batch": 8, "device": "cuda", "marker_placement": "ssm2_67", "body_models": {"male": h, "female": h}})
    return p, h, BodyModel(bm), mk


def _xb(B, seed):
    g = torch.Generator().manual_seed(seed)
    xb = torch.zeros(B, 93)
    xb[:, :3] = torch.randn(B, 3, generator=g)
    xb[:, 3:6] = torch.randn(B, 3, generator=g) * 0.9
    xb[:, 6:69] = torch.randn(B, 63, generator=g) * 0.25
    xb[:, 69:] = torch.randn(B, 24, generator=g) * 0.4
    return xb, torch.randn(10, generator=g)


def test_canonical_frame_kernel_matches_reference_golden():
    """egx_canonical_frame against CanonicalCoordinateExtractor.get_new_coordinate_torch's own outputs (canon_ref.npz)."""
    from egogen_amd import _lib
    lib = _lib.load()
    g = load_golden("canon_ref.npz")
    j = torch.from_numpy(g["jts"]).cuda().contiguous()
    B = j.shape[0]
    R, T = torch.empty(B, 3, 3, device="cuda"), torch.empty(B, 3, device="cuda")
    _lib.check(lib.egx_canonical_frame(_lib.ptr(j), int(j.shape[1]), B, _lib.ptr(R), _lib.ptr(T), _lib.current_stream_ptr()), "frame")
    assert max_abs(R.cpu(), g["R"]) < 1e-6
    assert max_abs(T.cpu(), g["T"].reshape(B, 3)) == 0.0
    with pytest.raises(_lib.EgxError):
        _lib.check(lib.egx_canonical_frame(_lib.ptr(j), 2, B, _lib.ptr(R), _lib.ptr(T), None), "frame")


def test_smplx_parser_get_new_coordinate_and_update_transl_glorot():
    """SMPLXParser.get_new_coordinate / calc_calibrate_offset / update_transl_glorot (baseops.py:465-598) vs the oracle."""
    from oracle import env as oenv
    from oracle.rot import tgm_angle_axis_to_rotation_matrix as aa2R
    from oracle.smplx_lbs import smplx_forward
    p, h, ob, mk = _parser()
    B = 9
    xb, betas = _xb(B, 4)
    _, j = smplx_forward(ob, xb, betas[None].repeat(B, 1))
    Ro, To = oenv.get_new_coordinate(j[:, :22])
    # tensors in, tensors out (to_numpy=False): [b,3,3], [b,1,3]
    R, T = p.get_new_coordinate(betas.cuda(), "male", xb.cuda(), to_numpy=False)
    assert R.shape == (B, 3, 3) and T.shape == (B, 1, 3) and R.is_cuda
    assert max_abs(R.cpu(), Ro) < 2e-5 and max_abs(T.cpu(), To) < 2e-5
    # numpy in, numpy out
    Rn, Tn = p.get_new_coordinate(betas.numpy(), "male", xb.numpy(), to_numpy=True)
    assert isinstance(Rn, np.ndarray) and max_abs(Rn, Ro) < 2e-5 and max_abs(Tn, To) < 2e-5
    # joints / markers getters share the forward (instantiation coverage of the pass-through methods)
    assert max_abs(p.get_jts(betas.cuda(), "male", xb.cuda(), to_numpy=False).cpu(), j[:, :22]) < 2e-5
    assert p.get_markers(betas.numpy(), "male", xb.numpy()).shape == (B, len(mk), 3)
    assert p.marker == [int(v) for v in mk]
    # calc_calibrate_offset: pelvis at zero orient / transl
    xz = xb.clone(); xz[:, :6] = 0
    _, jz = smplx_forward(ob, xz, betas[None].repeat(B, 1))
    d = p.calc_calibrate_offset(h, betas.cuda(), xb[:, 6:69].cuda(), to_numpy=False)
    assert max_abs(d.cpu(), jz[:, 0]) < 2e-5
    # update_transl_glorot into the frame of body 0 (one frame for all) and per-body frames
    for Rf, Tf in ((Ro[:1], To[:1]), (Ro, To)):
        ref = oenv.update_transl_glorot(Rf.expand(B, 3, 3), Tf.expand(B, 1, 3), jz[:, 0], xb)
        src = xb.clone().cuda()
        out = p.update_transl_glorot(Rf.cuda(), Tf.cuda(), betas.cuda(), "male", src, to_numpy=False, inplace=False)
        assert out.data_ptr() != src.data_ptr() and torch.equal(src.cpu(), xb)          # inplace=False leaves xb alone
        assert max_abs(out[:, :3].cpu(), ref[:, :3]) < 2e-5
        assert max_abs(aa2R(out[:, 3:6].cpu()), aa2R(ref[:, 3:6])) < 2e-5
        assert torch.equal(out[:, 6:].cpu(), xb[:, 6:])
        out2 = p.update_transl_glorot(Rf.cuda(), Tf.cuda(), betas.cuda(), "male", src, to_numpy=False, inplace=True)
        assert out2.data_ptr() == src.data_ptr() and max_abs(src[:, :6].cpu(), out[:, :6].cpu()) == 0.0
        arr = xb.numpy().copy()
        out3 = p.update_transl_glorot(Rf.numpy(), Tf.numpy(), betas.numpy(), "male", arr, to_numpy=True, inplace=True)
        assert out3 is arr and max_abs(arr[:, :6], out[:, :6].cpu()) < 1e-6
    # the transformed parameters describe the same body in the new frame: joints' = R^T (joints - T)
    out = p.update_transl_glorot(Ro[:1].cuda(), To[:1].cuda(), betas.cuda(), "male", xb.cuda(), to_numpy=False, inplace=False)
    j_new = p.get_jts(betas.cuda(), "male", out, to_numpy=False).cpu()
    want = torch.einsum("ij,bpj->bpi", Ro[0].T, j[:, :22] - To[0])
    assert max_abs(j_new, want) < 5e-5


def test_get_feature_kernel_matches_reference_golden():
    """egx_env_get_feature (the device functions of the step / reset kernels) against CrowdEnv._get_feature's own outputs."""
    from egogen_amd import _lib
    lib = _lib.load()
    g = load_golden("feature_ref.npz")
    c = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).cuda()
    nb, nt = g["pel"].shape[:2]
    dist = torch.empty(nb, nt, device="cuda")
    fea = torch.empty(nb, nt, 201, device="cuda")
    Y_l, pel, R0, wp = c("Y_l"), c("pel"), c("R0"), c("wpath")      # keep the device copies alive across the call
    T0 = c("T0").reshape(nb, 3).contiguous()
    _lib.check(lib.egx_env_get_feature(_lib.ptr(Y_l), _lib.ptr(pel), _lib.ptr(R0), _lib.ptr(T0), _lib.ptr(wp), 1,
                                       nb, nt, 67, _lib.ptr(dist), _lib.ptr(fea), _lib.current_stream_ptr()), "feature")
    torch.cuda.synchronize()
    assert max_abs(dist.cpu(), g["dist_xyz"].reshape(nb, nt)) < 1e-6
    # unit vectors; the marker sitting exactly on the target yields 0/1e-12 = 0 on both sides
    assert max_abs(fea.cpu(), g["fea_marker_3d_n"]) < 2e-6


def test_get_map_kernel_matches_reference_golden():
    """egx_env_get_map against batch_gen_amass.get_map / the box env's {1,-1} map (frames on the obstacle's edges)."""
    from egogen_amd import _lib
    lib = _lib.load()
    g = load_golden("getmap_ref.npz")
    tris = torch.from_numpy(g["tris"].reshape(-1, 6)).cuda().contiguous()
    R = torch.from_numpy(g["R"]).cuda().contiguous()
    T = torch.from_numpy(g["T"].reshape(-1, 3)).cuda().contiguous()
    nb = R.shape[0]
    lin = torch.linspace(-0.8, 0.8, 16).cuda()
    ps = torch.empty(nb, 256, 3, device="cuda")
    mp = torch.empty(nb, 256, device="cuda")
    _lib.check(lib.egx_env_get_map(_lib.ptr(tris), int(tris.shape[0]), float(g["floor_height"]), _lib.ptr(lin), 16, _lib.ptr(R), _lib.ptr(T), nb,
                                   _lib.ptr(ps), _lib.ptr(mp), _lib.current_stream_ptr()), "map")
    assert max_abs(ps.cpu(), g["points_scene"]) < 1e-6
    got, want = mp.cpu().numpy(), g["box_local_map"]
    # a grid point within round-off of a triangle edge may flip; everything else is exact
    diff = np.argwhere(got != want)
    for b, pidx in diff:
        px, py = g["points_scene"][b, pidx, :2]
        t = g["tris"]
        dmin = min(abs((px - t[f, (k + 1) % 3, 0]) * (t[f, k, 1] - t[f, (k + 1) % 3, 1]) - (t[f, k, 0] - t[f, (k + 1) % 3, 0]) * (py - t[f, (k + 1) % 3, 1]))
                   for f in range(t.shape[0]) for k in range(3))
        assert dmin < 1e-5, (b, pidx, dmin)
    assert len(diff) <= 2


# ---------------------------------------------------------------------------------------------
# BASELINE configs[0]: Replica room0
# ---------------------------------------------------------------------------------------------

def _room0_world(A, V=1536, sdf_res=48, finetuning=False, keep_rollout=False):
    """GPU VecCrowdEnv + CPU oracle on the REAL room0 walkable polygon / start-target pairs (egogen_assets.npz) with a
    room0-shaped synthetic SDF (room0_sdf.pkl is not redistributable)."""
    from egogen_amd.body_model import BodyModelHandle
    from egogen_amd.crowd_env import VecCrowdEnv
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG, VPoserEncoder
    from oracle.env import OracleCrowdEnv
    from oracle.smplx_lbs import BodyModel
    from tests.helpers import seeded_prior_state_dict, seeded_vposer_state_dict
    bm = synth.make_body_model(0, num_verts=V)
    mk, feet, fmi = synth.marker_ids(V), synth.feet_vids(V), synth.feet_marker_idx()
    prior_sd, vposer_sd = seeded_prior_state_dict(), seeded_vposer_state_dict()
    scene = synth.make_sdf_scene(sdf_res, room="room0")
    rings = synth.room0_polygon()
    assert len(rings) == 6 and sum(len(r) for r in rings) == 95
    pairs = np.asarray(synth.load_assets()["room0_pairs"][:256], np.float32)
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    o = OracleCrowdEnv(BodyModel(bm), prior_sd, {k: v.float() for k, v in vposer_sd.items()}, mk, feet, fmi, scene_kind="sdf",
                       sdf_dict=sd, edges=synth.rings_to_edges(rings), finetuning=finetuning)
    h = BodyModelHandle(bm, mk, feet)
    combo = GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)
    combo.load_state_dict(prior_sd)
    vp = VPoserEncoder()
    vp.load_state_dict(vposer_sd)
    env = VecCrowdEnv(A, h, combo.cuda().eval(), vp.cuda().eval(), scene_kind="sdf", sdf_dict=scene, rings=rings, pairs=pairs,
                      finetuning=finetuning, keep_rollout=keep_rollout, seed=0)
    return {"env": env, "oracle": o, "A": A, "pairs": pairs, "bm": bm}


def test_room0_reset_and_steps_match_oracle():
    """reset + 3 steps on room0: egosensing rays against 89 edges of a concave polygon with 5 holes."""
    from tests.test_env_gpu import _close, _compare_state, _oracle_reset, _sync_oracle_from_gpu, _world_scale
    A = 6
    w = _room0_world(A)
    env, o = w["env"], w["oracle"]
    assert env.edges.shape[0] == 89
    vp = env.valid_pairs[:A].cpu().numpy()
    env.set_candidates(vp.reshape(A, 1, 2, 3))
    obs = env.reset()
    oobs, accept = _oracle_reset(w, vp, [0] * A)
    assert bool(accept.all())
    _compare_state(w)
    _close(obs["egosensing"], oobs["egosensing"], 1e-4, "room0 reset egosensing", _world_scale(w))
    ego = obs["egosensing"].cpu().numpy()
    assert (ego < 0.999).any() and (ego > -0.999).any(), "rays should both hit walls and run free in room0"
    from oracle.env import calc_egosensing
    edges = synth.rings_to_edges(synth.room0_polygon())
    g = torch.Generator().manual_seed(11)
    for it in range(3):
        _sync_oracle_from_gpu(w)
        R0o, T0o = env.R0.clone(), env.T0.clone()
        z = torch.randn(A, 128, generator=g) * 0.5
        obs, rew, term = env.step(z.cuda(), auto_reset=False)
        oobs, orew, oterm = o.step(z)
        _close(env.Y_gen, o.last["Y_gen"], 1e-4, "Y_gen")
        _close(env.joints.reshape(A, 20, -1, 3), o.last["joints"], 1e-4, "joints")
        # (1) the ray caster itself: the oracle's restatement of _calc_egosensing evaluated on the GPU's OWN world joints of
        #     the new seed frames - isolates E4 on the 89-edge polygon from upstream round-off
        jw = torch.einsum("bij,btpj->btpi", R0o, env.joints.reshape(A, 20, -1, 3)[:, 18:20]) + T0o[:, None, None, :]
        ego_ref = torch.stack([calc_egosensing(jw[a].cpu(), edges) for a in range(A)])
        # the look-at direction is a difference of world-frame joints (metres, fp32: ~5e-7 each); the kernel forms them in
        # its own operation order, so the direction agrees to ~2e-6 / |look_xy| rad and a hit at a few metres moves by that
        # times the distance: tolerance per (agent, frame) from the length of ITS look vector; the typical ray is exact
        look = (jw[:, :, 57] - jw[:, :, 23] + jw[:, :, 56] - jw[:, :, 24])[..., :2].norm(dim=-1).cpu()        # [A,2]
        d1 = (obs["egosensing"].cpu() - ego_ref).abs()                                                       # [A,2,32]
        tol1 = 2e-5 + 2e-5 / look.clamp(min=1e-3)
        assert (d1 <= tol1[..., None]).all(), (float(d1.max()), float(look.min()))
        assert float(d1.median()) <= 5e-6
        # (2) end to end against the oracle's own joints: a 2e-5 m joint difference turns the look-at direction (a ~6 cm
        #     baseline between the eye joints) by up to ~3e-4 rad, i.e. millimetres at a wall seen obliquely 7 m away, and
        #     a ray grazing a polygon corner may switch edges - so: nearly all rays at 2e-4, every ray at 1e-2
        d = (obs["egosensing"].cpu() - oobs["egosensing"]).abs()
        assert float((d <= 2e-4).float().mean()) >= 0.9 and float(d.max()) <= 1e-2, (float((d <= 2e-4).float().mean()), float(d.max()))
        _close(rew, orew, 1e-4, "reward")
        assert term.cpu().bool().tolist() == oterm.tolist()
        _compare_state(w)


def test_single_agent_crowd_env_view_rollout_matches_oracle(tmp_path):
    """configs[0] shape: ONE agent behind the reference's gym interface (CrowdEnv.reset / step, crowd_env_2f.py:78,320) on
    room0, actions injected, the episode written with save_rollout_results (crowd_env_2f.py:154-155,305-309) and compared
    with an oracle rollout on the same actions: blended_marker, smplx_params, pelvis_loc, transf_* at 1e-4."""
    from egogen_amd.crowd_env import CrowdEnv
    from egogen_amd.utils import save_rollout_results
    from tests.test_env_gpu import _oracle_reset
    w = _room0_world(1, keep_rollout=True)
    vec, o = w["env"], w["oracle"]
    env = CrowdEnv(vec)
    assert env.action_space.shape == (128,) and env.observation_space["state"].shape == (2, 402)
    pair = vec.valid_pairs[3:4].cpu().numpy()
    vec.set_candidates(pair.reshape(1, 1, 2, 3))
    obs, info = env.reset()
    assert info == {} and obs["state"].shape == (2, 402) and obs["egosensing"].shape == (2, 32)
    assert obs["dist"].shape == (1,) and obs["time"].shape == (1,)
    oobs, _ = _oracle_reset(w, pair, [0])
    assert max_abs(obs["state"].cpu(), oobs["state"][0]) < 1e-4
    g = torch.Generator().manual_seed(5)
    mps, ref = [], []
    wpath0 = vec.wpath[0].clone()
    for it in range(4):                        # no re-synchronisation: the oracle runs its own state for the whole episode
        z = (torch.randn(128, generator=g) * 0.5).numpy().astype(np.float32)      # the reference passes numpy (crowd_env_2f.py:102)
        R_prev, T_prev = o.R0.clone(), o.T0.clone()
        obs, rew, term, trunc, info = env.step(z)
        oobs, orew, oterm = o.step(torch.from_numpy(z)[None])
        assert isinstance(rew, float) and isinstance(term, bool) and trunc is False and info == {}
        assert abs(rew - float(orew[0])) < 3e-3 and term == bool(oterm[0])
        pel = vec.joints.reshape(1, 20, -1, 3)[:, :, 0]
        fr = vec.prev_frame[0]
        mps.append([vec.marker_b[0:1].clone(), vec.pred_params[0:1].clone(), vec.betas[0].clone(), "male", fr[:9].reshape(3, 3).clone(),
                    fr[9:].reshape(1, 3).clone(), pel.clone(), "2-frame"])
        ref.append({"blended_marker": o.last["marker_b"][0].numpy(), "smplx_params": o.last["pred_params"][0:1].numpy(),
                    "pelvis_loc": o.last["pelvis"][0].numpy(), "transf_rotmat": R_prev[0].numpy(), "transf_transl": T_prev[0].numpy()})
        if term:
            break
    path = save_rollout_results({"wpath": wpath0, "navmesh_path": "room0"}, mps, str(tmp_path), man_id="view")
    with open(path, "rb") as f:
        node = pickle.load(f)
    assert list(node.keys()) == ["motion", "wpath", "navmesh_path"] and len(node["motion"]) == len(ref)
    drift = []
    for mp, r in zip(node["motion"], ref):
        assert mp["blended_marker"].shape == (20, 67, 3) and mp["smplx_params"].shape == (1, 20, 93) and mp["pelvis_loc"].shape == (20, 3)
        assert mp["gender"] == "male" and mp["mp_type"] == "2-frame" and mp["betas"].shape == (10,)
        for k in ("blended_marker", "pelvis_loc", "transf_rotmat", "transf_transl"):
            scale = max(1.0, float(np.abs(r[k]).max()))
            assert max_abs(mp[k], r[k]) <= 1e-4 * scale, (k, max_abs(mp[k], r[k]))
        assert max_abs(mp["smplx_params"][..., :3], r["smplx_params"][..., :3]) <= 2e-4
        assert max_abs(mp["smplx_params"][..., 6:], r["smplx_params"][..., 6:]) <= 2e-4
        drift.append(max_abs(mp["blended_marker"], r["blended_marker"]))
    print("un-synchronised marker drift per primitive:", ["%.2e" % d for d in drift])


def test_rotmat_to_angle_axis_device_function_matches_in_tree_copy_values():
    """`egx_tgm_rotmat_to_aa` (the R -> axis-angle tail of the regressor, of reset and of `update_transl_glorot`) through the
    C ABI: with the identity frame `egx_update_transl_glorot` returns log(exp(glorot)), compared with the golden of the
    reference tree's own rotation_matrix_to_angle_axis (tests/golden/rot2aa_ref.npz, scripts/gen_goldens.py rot2aa)."""
    from egogen_amd import _lib
    from tests.helpers import load_golden
    lib = _lib.load()
    g = load_golden("rot2aa_ref.npz")
    n = g["aa_in"].shape[0]
    xb = torch.zeros(n, 93)
    xb[:, 3:6] = torch.from_numpy(g["aa_in"])
    xb = xb.cuda()
    R, T, delta = torch.eye(3).reshape(1, 3, 3).cuda(), torch.zeros(1, 3).cuda(), torch.zeros(n, 3).cuda()
    out = torch.empty_like(xb)
    _lib.check(lib.egx_update_transl_glorot(_lib.ptr(R), _lib.ptr(T), 1, _lib.ptr(delta), _lib.ptr(xb), n, _lib.ptr(out),
                                            _lib.current_stream_ptr()), "egx_update_transl_glorot")
    theta = np.linalg.norm(g["aa_in"], axis=1)
    err = np.abs(out[:, 3:6].cpu().numpy() - g["aa_out"]).max(axis=1)
    assert err[theta < 2.0].max() < 1e-6 and err.max() < 5e-6, (err[theta < 2.0].max(), err.max())


def test_crowd_env_behind_a_sequential_vector_env_loop():
    """The per-agent `CrowdEnv` is what tianshou's `DummyVectorEnv` / `Collector` hold (main_ppo.py:97-101; the reference's own
    copy: crowd_ppo/dummy_vector_env.py:81-128): they read `action_space.shape / .sample() / .contains()`, iterate
    `observation_space.spaces`, send one action per env, collect (obs, rew, terminated, truncated, info) tuples, add `env_id`
    to the info dict, `np.stack` the pieces and reset finished envs by id.  The same loop, written out, over two views."""
    from egogen_amd.crowd_env import CrowdEnv
    worlds = [_room0_world(1), _room0_world(1)]
    envs = [CrowdEnv(w["env"]) for w in worlds]
    sp = envs[0].action_space
    assert sp.shape == (128,) and float(sp.low.min()) == -6.0 and float(sp.high.max()) == 6.0 and sp.dtype == np.float32
    osp = envs[0].observation_space
    assert list(osp.spaces.keys()) == ["state", "egosensing", "dist", "time"]
    assert osp["state"].shape == (2, 402) and osp["egosensing"].shape == (2, 32) and osp["dist"].shape == (1,) and osp["time"].shape == (1,)
    assert (float(osp["state"].low.min()), float(osp["state"].high.max())) == (-2.0, 2.0) and float(osp["dist"].low.min()) == 0.0
    sp.seed(0)
    a0 = sp.sample()
    assert a0.shape == (128,) and a0.dtype == np.float32 and sp.contains(a0) and not sp.contains(np.full(128, 7.0, np.float32))

    def to_np(obs):
        return {k: v.detach().cpu().numpy() for k, v in obs.items()}

    # DummyVectorEnv.reset: every worker resets, observations stacked
    ret = [e.reset(seed=11 + i) for i, e in enumerate(envs)]
    obs_list = [to_np(r[0]) for r in ret]
    assert all(r[1] == {} for r in ret)
    for o in obs_list:      # what a Collector would put into its buffer lies in the declared spaces, up to the egosensing scale
        assert set(o) == set(osp.spaces) and all(o[k].shape == osp[k].shape for k in o)
        assert osp["dist"].contains(o["dist"]) and osp["time"].contains(o["time"]) and osp["state"].contains(o["state"])
    ids = list(range(len(envs)))
    n_done = 0
    for step in range(16):
        action = np.stack([sp.sample() * 0.1 for _ in ids])      # policy output, one row per ready env
        assert len(action) == len(ids)
        result = []
        for i, j in enumerate(ids):                               # workers[j].send(action[i]) ... recv()
            env_return = list(envs[j].step(action[i]))
            env_return[-1]["env_id"] = j
            result.append(env_return)
        obs_l, rew_l, term_l, trunc_l, info_l = tuple(zip(*result))
        obs_stack = np.stack([to_np(o)["state"] for o in obs_l])
        rew, term, trunc, info = np.stack(rew_l), np.stack(term_l), np.stack(trunc_l), np.stack(info_l)
        assert obs_stack.shape == (2, 2, 402) and rew.shape == (2,) and rew.dtype == np.float64
        assert term.dtype == bool and trunc.dtype == bool and not trunc.any() and [d["env_id"] for d in info] == ids
        assert np.isfinite(rew).all()
        done = np.logical_or(term, trunc)
        for j in np.where(done)[0]:                               # Collector: reset the finished envs by id
            o, inf = envs[j].reset()
            assert inf == {} and to_np(o)["time"].tolist() == [1.0]
            n_done += 1
    assert n_done >= 1, "max_depth 13 must end at least one episode within 16 steps"


def test_rollout_primitives_matches_reference_restatement():
    """egogen_amd.utils.rollout_primitives (vis.py:44-78, the consumer of motion_*.pkl) against the oracle's
    line-by-line restatement (scipy rotations like the reference), and the defining property: the body posed with the
    rolled-out parameters IS the canonical body carried into the world frame."""
    from egogen_amd.utils import rollout_primitives
    from oracle.rollout import rollout_primitives as oracle_rollout
    from oracle.smplx_lbs import smplx_forward
    from scipy.spatial.transform import Rotation
    p, h, ob, mk = _parser()
    g = torch.Generator().manual_seed(8)
    betas = torch.randn(10, generator=g)
    mps = []
    for i in range(3):
        xb = torch.zeros(20, 93)
        xb[:, :3] = torch.randn(20, 3, generator=g) * 0.5
        xb[:, 3:6] = torch.randn(20, 3, generator=g) * 0.7
        xb[:, 6:69] = torch.randn(20, 63, generator=g) * 0.2
        R = torch.from_numpy(Rotation.from_euler("z", float(torch.rand(1, generator=g)) * 6.28).as_matrix()).float()
        mps.append({"smplx_params": xb[None].numpy(), "betas": betas.numpy(), "gender": "male", "transf_rotmat": R.numpy(),
                    "transf_transl": (torch.randn(1, 3, generator=g) * 2).numpy(), "mp_type": "2-frame",
                    "blended_marker": np.zeros((20, 67, 3), np.float32), "pelvis_loc": np.zeros((20, 3), np.float32)})

    def pelvis_of(b):
        _, j = smplx_forward(ob, torch.zeros(1, 93), torch.as_tensor(b).reshape(1, 10).float())
        return j[0, 0].numpy()
    got = rollout_primitives(mps, h)
    ref = oracle_rollout(mps, pelvis_of)
    assert got.shape == ref.shape == (20 + 18 + 18, 93)
    assert max_abs(got[:, :3], ref[:, :3]) < 2e-5 and max_abs(got[:, 6:], ref[:, 6:]) == 0.0
    assert max_abs(Rotation.from_rotvec(got[:, 3:6]).as_matrix(), Rotation.from_rotvec(ref[:, 3:6]).as_matrix()) < 2e-5
    # property: joints(rolled-out params) == R joints(canonical params) + T for the first primitive
    j_w = p.get_jts(betas.cuda(), "male", torch.from_numpy(got[:20]).cuda(), to_numpy=False).cpu()
    j_c = p.get_jts(betas.cuda(), "male", torch.from_numpy(mps[0]["smplx_params"][0]).cuda(), to_numpy=False).cpu()
    want = torch.einsum("ij,bpj->bpi", torch.from_numpy(mps[0]["transf_rotmat"]), j_c) + torch.from_numpy(mps[0]["transf_transl"])
    assert max_abs(j_w, want) < 5e-5


def test_main_ppo_watch_room0_one_agent(tmp_path):
    """configs[0] command line: main_ppo.py --watch --test-num 1 on room0 writes motion_*.pkl and config.yaml."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "crowd_ppo", "main_ppo.py"), "--watch", "--deterministic-eval", "--test-num", "1",
           "--scene", "room0", "--num-verts", "1024", "--sdf-res", "32", "--logdir", str(tmp_path / "log")]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "Final reward:" in r.stdout
    assert os.path.isfile(tmp_path / "results" / "crowd_ppo" / "MPVAEPolicy_samp_collision" / "collision_test" / "config.yaml")
    pk = sorted((tmp_path / "log" / "eval_results").glob("motion_*.pkl"))
    assert pk, "no rollout written"
    with open(pk[0], "rb") as f:
        node = pickle.load(f)
    assert node["wpath"].shape == (2, 3) and node["motion"][0]["blended_marker"].shape == (20, 67, 3)


def test_canonicalize_samp_matches_reference_golden(tmp_path):
    """exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py on the GPU operators (get_new_coordinate, update_transl_glorot,
    SMPL-X forward) against the outputs of the reference's own script on the same synthetic SAMP pickle
    (tests/golden/canonicalize_ref.npz, body model = synthetic full size), then the file loop: names, keys, dtypes, and that the
    marker-predictor's batch generator reads what was written."""
    import pickle
    from egogen_amd.body_model import BodyModelHandle, SMPLXParser
    from egogen_amd.canonicalize import canonicalize_frames, canonicalize_samp
    from egogen_amd.train_predictor import BatchGeneratorAMASSCanonicalized
    from tests.helpers import load_golden
    g = load_golden("canonicalize_ref.npz")
    h = BodyModelHandle(synth.make_body_model(int(g["body_model_seed"])), synth.marker_ids(), synth.feet_vids())
    parser = SMPLXParser({"n_batch": 20, "device": "cuda", "marker_placement": "ssm2_67", "body_models": {"male": h}})
    for i, s in enumerate((0, 60)):
        d = canonicalize_frames(parser, g["in_trans"][s:s + 60:3], g["in_poses"][s:s + 60:3], g["in_betas"])
        for k, tol in (("transf_rotmat", 1e-5), ("transf_transl", 1e-5), ("trans", 1e-4), ("betas", 0), ("joints", 1e-4),
                       ("marker_cmu_41", 1e-4), ("marker_ssm2_67", 1e-4)):
            ref = g[f"out{i}_{k}"]
            assert d[k].shape == ref.shape and d[k].dtype == ref.dtype, (k, d[k].dtype, ref.dtype)
            assert np.abs(d[k].astype(np.float64) - ref).max() <= tol * max(1.0, np.abs(ref).max()), (i, k)
        # axis-angle vectors are compared as rotations (the reference goes through scipy, the kernel through torchgeometry's formulas)
        from oracle.rot import tgm_angle_axis_to_rotation_matrix as aa2R
        Ra, Rb = aa2R(torch.tensor(d["poses"][:, :3], dtype=torch.float64)), aa2R(torch.tensor(g[f"out{i}_poses"][:, :3], dtype=torch.float64))
        assert float((Ra - Rb).abs().max()) < 1e-5
        assert np.array_equal(d["poses"][:, 3:], g[f"out{i}_poses"][:, 3:]) and d["poses"].shape == (20, 165)
    # the file loop of the script's __main__ (:192-290)
    root = tmp_path / "samp"
    root.mkdir()
    for name, n in (("locomotion_a_stageII.pkl", 127), ("locomotion_b_stageII.pkl", 50), ("run_c_stageII.pkl", 190)):
        k = min(n, len(g["in_trans"]))
        reps = -(-n // k)
        with open(root / name, "wb") as f:
            pickle.dump({"mocap_framerate": 120.0, "pose_est_trans": np.tile(g["in_trans"][:k], (reps, 1))[:n],
                         "pose_est_fullposes": np.tile(g["in_poses"][:k], (reps, 1))[:n], "shape_est_betas": g["in_betas"]}, f)
    counts = canonicalize_samp(parser, 1, str(root), verbose=False)
    assert counts["locomotion"] == 2 and counts["run"] == 3 and counts["chair"] == 0      # 43 -> 2, 17 -> skipped, 64 -> 3
    files = sorted((root / "Canonicalized-MP" / "data" / "run").glob("subseq_*.npz"))
    assert [f.name for f in files] == ["subseq_00000.npz", "subseq_00001.npz", "subseq_00002.npz"]
    with np.load(files[0]) as z:
        assert set(z.files) == {"transf_rotmat", "transf_transl", "trans", "poses", "betas", "gender", "mocap_framerate", "joints",
                                "marker_cmu_41", "marker_ssm2_67"}                      # the keys of data/locomotion/subseq_00343.npz
        assert z["joints"].dtype == np.float32 and z["trans"].dtype == np.float64 and z["transf_transl"].shape == (1, 3)
    loco = sorted((root / "Canonicalized-MP" / "data" / "locomotion").glob("*.npz"))
    with np.load(loco[0]) as z:
        assert np.abs(z["marker_ssm2_67"] - g["out0_marker_ssm2_67"]).max() < 1e-4       # first file = frames 0, 3, .., 57 of sequence a
    gen = BatchGeneratorAMASSCanonicalized(str(root / "Canonicalized-MP" / "data"), amass_subset_name=["locomotion", "run"], sample_rate=1)
    gen.get_rec_list(shuffle_seed=0)
    assert gen.data_all.shape[0] == 5 and gen.data_all.shape[2] == 201
