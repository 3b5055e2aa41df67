"""Pin the CPU oracle against vectors produced by the reference's own code
(scripts/gen_goldens.py imported the reference classes in the build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import nets, rot, sdf
from tests.helpers import load_golden, max_abs, rebuild_state_dict


def test_calc_sdf_matches_reference():
    g = load_golden("calc_sdf_ref.npz")
    for tag in "abc":
        d = {"sdf": torch.from_numpy(g[f"{tag}_sdf"]), "center": torch.from_numpy(g[f"{tag}_center"]),
             "scale": torch.from_numpy(g[f"{tag}_scale"])}
        out = sdf.calc_sdf(torch.from_numpy(g[f"{tag}_pts"]), d).numpy()
        assert out.shape == g[f"{tag}_val"].shape
        assert max_abs(out, g[f"{tag}_val"]) < 2e-6, tag


def test_cvae_decode_matches_reference():
    g = load_golden("cvae_ref.npz")
    sd = rebuild_state_dict(g, [g["fill_seed"]], [""])
    sd = {"predictor." + k: v for k, v in sd.items()}
    Y = nets.cvae_decode(sd, torch.from_numpy(g["X"]), torch.from_numpy(g["z"])).numpy()
    assert Y.shape == (18, 5, 201)
    assert max_abs(Y, g["Y"]) < 2e-5


def test_regressor_6d_and_cont2rotmat_match_reference():
    g = load_golden("regressor_ref.npz")
    sd = rebuild_state_dict(g, [g["fill_seed"]], [""], gains=[float(g["fill_gain"])])
    sd = {"regressor." + k: v for k, v in sd.items()}
    xb6 = nets.regressor_6d(sd, torch.from_numpy(g["markers"]), torch.from_numpy(g["betas"]))
    assert max_abs(xb6.numpy(), g["xb6"]) < 2e-5 * max(1.0, np.abs(g["xb6"]).max())
    R = rot.cont2rotmat(torch.from_numpy(g["xb6"])[:, 3:135].contiguous().view(7, -1, 6)).numpy()
    assert max_abs(R, g["rotmat"]) < 1e-6
    # full forward: 6D -> axis-angle through the restated tgm path must reproduce the rotation
    xb = nets.regressor_forward(sd, torch.from_numpy(g["markers"]), torch.from_numpy(g["betas"]))
    assert xb.shape == (7, 93)
    Rback = rot.tgm_angle_axis_to_rotation_matrix(xb[:, 3:69].reshape(-1, 3)).numpy()
    assert max_abs(Rback, g["rotmat"]) < 5e-5


def test_policy_matches_reference():
    g = load_golden("policy_ref.npz")
    sd = rebuild_state_dict(g, g["fill_seeds"], ["shared_net.", "actor.", "critic."], gains=[1.0, 1.4, 1.4])
    assert len(sd) == 28
    obs = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("obs_")}
    hx = nets.policy_base(sd, obs)
    mu, logvar = nets.policy_actor(sd, hx)
    val = nets.policy_critic(sd, hx)
    assert max_abs(hx.numpy(), g["hx"]) < 2e-5
    assert max_abs(mu.numpy(), g["mu"]) < 1e-4 * max(1.0, np.abs(g["mu"]).max())
    assert max_abs(logvar.numpy(), g["logvar"]) < 1e-4 * max(1.0, np.abs(g["logvar"]).max())
    assert max_abs(val.numpy(), g["value"]) < 1e-4 * max(1.0, np.abs(g["value"]).max())


def test_canonical_frame_matches_reference():
    from oracle import env as oenv
    g = load_golden("canon_ref.npz")
    R, T = oenv.get_new_coordinate(torch.from_numpy(g["jts"]))
    assert max_abs(R.numpy(), g["R"]) < 1e-6
    assert max_abs(T.numpy(), g["T"]) == 0.0


def test_angle_axis_to_rotation_matrix_matches_in_tree_copy():
    """rot.tgm_angle_axis_to_rotation_matrix against the outputs of the reference tree's kornia-derived
    angle_axis_to_rotation_matrix (experiments/HMR/prohmr/utils/konia_transform.py:234-310; same formulas and the same
    theta^2 > 1e-6 switch as torchgeometry 0.1.2), angles from 1e-5 to 3 rad and the zero vector."""
    from oracle import rot
    g = load_golden("aa2rot_ref.npz")
    R = rot.tgm_angle_axis_to_rotation_matrix(torch.from_numpy(g["aa"]))
    assert max_abs(R.numpy(), g["R"]) < 1e-6
    # and the HIP-side convention (egx_tgm_aa_to_rotmat mirrors the same function) is covered by tests/test_env_gpu.py


def test_rotation_matrix_to_angle_axis_matches_in_tree_copy_values():
    """rot.tgm_rotation_matrix_to_angle_axis (torchgeometry 0.1.2 restated: R -> quaternion by the four-mask rule -> axis-angle
    by 2 atan2) against the VALUES of the reference tree's kornia-derived rotation_matrix_to_angle_axis
    (experiments/HMR/prohmr/utils/konia_transform.py:316-341), a later algorithm of the same function: 256 rotations with
    theta from 1e-3 to 3.0 rad, all four quaternion branches populated.  Away from theta = pi the rotation vector is unique,
    so the two must agree to fp32 round-off (the conditioning of the inverse grows like 1 / (pi - theta): 3e-6 at 3 rad)."""
    from oracle import rot
    g = load_golden("rot2aa_ref.npz")
    assert np.bincount(g["branch"], minlength=4).min() >= 8          # every branch of the quaternion conversion is exercised
    out = rot.tgm_rotation_matrix_to_angle_axis(torch.from_numpy(g["R"])).numpy()
    theta = np.linalg.norm(g["aa_in"], axis=1)
    err = np.abs(out - g["aa_out"]).max(axis=1)
    assert err[theta < 2.0].max() < 1e-6 and err.max() < 3e-6, (err[theta < 2.0].max(), err.max())
    # float64 evaluation of the restatement: closer still to the exact rotation vector than either fp32 algorithm
    out64 = rot.tgm_rotation_matrix_to_angle_axis(torch.from_numpy(g["R"]).double()).numpy()
    assert np.abs(out64 - g["aa_in"]).max() < 3e-6


def test_other_rotation_restatements_agree_with_in_tree_values():
    """smplx's batch_rodrigues (angle = ||aa + 1e-8||, full Rodrigues formula) and pytorch3d's axis_angle_to_matrix /
    matrix_to_axis_angle (through quaternions) are restated from packages that are absent; each computes the same mathematical
    function as the reference tree's kornia-derived pair (experiments/HMR/prohmr/utils/konia_transform.py), whose VALUES the
    fixtures aa2rot_ref.npz / rot2aa_ref.npz hold (angles from 1e-5 to 3 rad; the small-angle branches differ by O(theta^2)
    <= 1e-10 where they apply)."""
    from oracle import rot
    g = load_golden("aa2rot_ref.npz")
    aa = torch.from_numpy(g["aa"])
    # the kornia copy divides by (theta + 1e-6): 1e-6 off the exact matrix around theta ~ 1e-3, which bounds the agreement
    assert max_abs(rot.smplx_batch_rodrigues(aa).numpy(), g["R"]) < 2e-6
    assert max_abs(rot.p3d_axis_angle_to_matrix(aa).numpy(), g["R"]) < 2e-6
    g2 = load_golden("rot2aa_ref.npz")
    # pytorch3d 0.7.4 does not standardise the quaternion's sign: where the selected candidate has a negative real part the
    # rotation vector comes out with an angle in (pi, 2 pi) - the same rotation, another representative - so the comparison is
    # between the rotations, not the vectors
    out = rot.p3d_matrix_to_axis_angle(torch.from_numpy(g2["R"]))
    Ra, Rb = rot.p3d_axis_angle_to_matrix(out), rot.p3d_axis_angle_to_matrix(torch.from_numpy(g2["aa_out"]))
    assert max_abs(Ra.numpy(), Rb.numpy()) < 2e-6
    assert max_abs(Ra.numpy(), g2["R"]) < 2e-6


def test_smplx_skinning_matches_in_tree_restatement():
    """oracle.smplx_lbs: pose-feature layout, pose-corrective product and linear blend skinning against the reference tree's
    own restatement of that tail, experiments/HOOD/utils/lbs.py::pose_garment (:85-124), run on the oracle's joint transforms
    (scripts/gen_goldens.py lbs_skin).  What stays restated from smplx 0.1.28: batch_rodrigues, batch_rigid_transform, the
    landmark path."""
    from egogen_amd import synth
    from oracle.smplx_lbs import BodyModel, smplx_forward
    g = load_golden("lbs_skin_ref.npz")
    ob = BodyModel(synth.make_body_model(int(g["body_seed"]), num_verts=int(g["num_verts"])))
    verts, _ = smplx_forward(ob, torch.from_numpy(g["xb"]), torch.from_numpy(g["betas"]))
    # V = 2048 (64 vertex tiles), 257 bodies (one past a 256-body group), translated; 220 vertices of each body are kept
    assert g["xb"].shape[0] >= 257 and int(g["num_verts"]) >= 2048 and np.abs(g["xb"][:, :3]).max() > 1.0
    assert max_abs(verts.numpy()[:, g["vertex_ids"]], g["verts"]) == 0.0


def test_get_feature_and_blend_params_match_reference():
    """oracle.env.get_feature / blend_params against crowd_env_2f.CrowdEnv._get_feature / _blend_params run unbound
    (scripts/gen_goldens.py); the fixture holds a marker and a pelvis exactly on the target (the clip(min=1e-12) branch)."""
    from oracle import env as oenv
    g = load_golden("feature_ref.npz")
    t = lambda k: torch.from_numpy(g[k])
    nb = g["pel"].shape[0]
    outs = oenv.get_feature(t("Y_l"), t("pel"), t("R0"), t("T0"), t("wpath")[None].repeat(nb, 1, 1))
    for o, k in zip(outs, ("dist_xy", "dist_xyz", "fea_wpath", "fea_marker_3d_n", "fea_marker_h")):
        assert max_abs(o.numpy(), g[k]) == 0.0, k
    assert float(g["dist_xyz"].min()) == pytest.approx(1e-12)
    out = oenv.blend_params(t("blend_in").clone(), 2)
    assert max_abs(out.numpy(), g["blend_out"]) == 0.0
    assert max_abs(out.numpy()[:, :, :6], g["blend_in"][:, :, :6]) == 0.0


def test_get_map_matches_reference():
    """oracle.env.get_map against batch_gen_amass.get_map and the box env's _get_feature map tail (crowd_env_2f_box.py:762-770),
    frames placed on the obstacle's edges / the floor's border so that both map values occur."""
    from oracle import env as oenv
    g = load_golden("getmap_ref.npz")
    pts, ps, inside = oenv.get_map(torch.from_numpy(g["tris"]), torch.from_numpy(g["R"]), torch.from_numpy(g["T"]), 16, 0.8,
                                   float(g["floor_height"]))
    assert max_abs(pts.numpy(), g["points"]) == 0.0
    assert max_abs(ps.numpy(), g["points_scene"]) < 1e-6
    assert np.array_equal(inside.numpy(), g["map"])
    assert 0 < int(g["map"].sum()) < g["map"].size
    assert np.array_equal(np.where(inside.numpy(), 1.0, -1.0), g["box_local_map"])
    assert max_abs(pts.numpy(), g["box_points_local"]) == 0.0


def _equal_tree(a, b, path=""):
    assert type(a) is type(b), (path, type(a), type(b))
    if isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()), (path, list(a.keys()), list(b.keys()))
        for k in a:
            _equal_tree(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _equal_tree(x, y, f"{path}[{i}]")
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), path
    else:
        assert a == b, path


def test_save_rollout_results_matches_reference(tmp_path):
    """egogen_amd.utils.save_rollout_results writes, for the same inputs, the very pickle payload the reference's
    crowd_ppo/utils.py::save_rollout_results wrote (tests/golden/rollout_ref.pkl): keys in the same order, same dtypes /
    shapes / values (this is what vis.py and gen_egobody_* read)."""
    import pickle
    from egogen_amd.utils import save_rollout_results
    from tests.helpers import GOLDEN
    g = load_golden("rollout_ref.npz")
    with open(os.path.join(GOLDEN, "rollout_ref.pkl"), "rb") as f:
        ref = pickle.load(f)
    mps = []
    for i in range(int(g["n_mp"])):
        t = lambda j: torch.from_numpy(g[f"mp{i}_{j}"])
        mps.append([t(0), t(1), t(2), "male", t(4), t(5), t(6), "2-frame"])
    scene = {"wpath": torch.from_numpy(g["wpath"]), "navmesh_path": ref["navmesh_path"], "scene_path": ref["scene_path"]}
    path = save_rollout_results(scene, mps, str(tmp_path / "out"), man_id="ref")
    assert os.path.basename(path) == "motion_ref.pkl"
    with open(path, "rb") as f:
        got = pickle.load(f)
    _equal_tree(got, ref)
    # the time-stamped name of the default call (utils.py:43-46)
    p2 = save_rollout_results(scene, mps[:1], str(tmp_path / "out"))
    assert os.path.basename(p2).startswith("motion_") and p2.endswith(".pkl") and p2 != path


def test_predictor_training_loss_and_gradients_match_reference():
    """oracle.train (C-VAE forward with encoder, reconstruction + temporal-difference + robust KL loss, the multi-primitive
    roll-out loss with on-the-fly re-canonicalisation) against GAMMAPrimitiveVAETrainOP.calc_loss / calc_loss_rollout of the
    reference run on the CPU with the same seeded weights, data and reparameterisation noise: loss terms and the gradient
    of every parameter (norm + first entries)."""
    from oracle import train as otrain
    g = load_golden("predictor_train_ref.npz")
    sd = rebuild_state_dict(g, [int(g["fill_seed"])], [""], gains=[float(g["fill_gain"])])
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    keys = [str(k) for k in g["grad_keys"]]
    loss, info = otrain.predictor_loss(sd, torch.from_numpy(g["data"]), torch.from_numpy(g["eps"]))
    np.testing.assert_allclose([float(v) for v in info], g["loss_info"], rtol=2e-6, atol=1e-7)
    grads = torch.autograd.grad(loss, [sd[k] for k in keys])
    for k, gr, n, hd in zip(keys, grads, g["grad_norm"], g["grad_head"]):
        assert abs(float(gr.norm()) - n) <= 2e-5 * max(n, 1e-6) + 1e-9, k
        assert max_abs(np.resize(gr.flatten()[:8].numpy(), 8), hd) <= 2e-5 * max(float(np.abs(hd).max()), 1e-4), k
    loss2, info2 = otrain.predictor_loss_rollout(sd, torch.from_numpy(g["roll_markers"]), torch.from_numpy(g["roll_jts"]),
                                                 [torch.from_numpy(e) for e in g["roll_eps"]])
    np.testing.assert_allclose(info2.numpy(), g["roll_loss_info"], rtol=2e-6, atol=1e-7)
    grads2 = torch.autograd.grad(loss2, [sd[k] for k in keys])
    for k, gr, n in zip(keys, grads2, g["roll_grad_norm"]):
        assert abs(float(gr.norm()) - n) <= 5e-5 * max(n, 1e-6) + 1e-9, k


def test_canonicalize_restatement_matches_the_reference_script():
    """oracle/canonicalize.py against what the reference's own canonicalize_subsequence (utils_canonicalize_samp.py:123-189)
    returned for two back-to-back 20-frame sub-sequences of a synthetic SAMP pickle (scripts/gen_goldens.py canonicalize)."""
    from egogen_amd import synth
    from oracle.canonicalize import canonicalize_frames
    from oracle.smplx_lbs import BodyModel
    g = load_golden("canonicalize_ref.npz")
    bm = BodyModel(synth.make_body_model(int(g["body_model_seed"])))
    assert np.array_equal(g["cmu_ids"], synth.load_assets()["cmu_marker_ids"]) and np.array_equal(g["ssm_ids"], synth.marker_ids())
    for i, s in enumerate((0, 60)):
        d = canonicalize_frames(bm, g["in_trans"][s:s + 60:3], g["in_poses"][s:s + 60:3], g["in_betas"], g["cmu_ids"], g["ssm_ids"])
        for k in ("transf_rotmat", "transf_transl", "trans", "poses", "betas", "joints", "marker_cmu_41", "marker_ssm2_67"):
            ref = g[f"out{i}_{k}"]
            assert np.asarray(d[k]).shape == ref.shape, k
            assert np.abs(np.asarray(d[k], np.float64) - ref).max() < 2e-6, (i, k)
        assert str(g[f"out{i}_gender"]) == "male" and int(g[f"out{i}_mocap_framerate"]) == 120
    # in its own frame the first body stands at the origin facing +y: pelvis xy = 0 up to the offset correction, hips along x
    j = g["out0_joints"][0]
    assert abs(j[0, 0]) < 1e-5 and abs(j[0, 1]) < 1e-5 and abs(j[2, 1] - j[1, 1]) < 1e-5 and j[2, 0] > j[1, 0]
