"""Invariants of the CPU oracle where the reference pins nothing (third-party algorithms)."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from egogen_amd import synth
from oracle import rot
from oracle.smplx_lbs import BodyModel, smplx_forward


def test_tgm_aa_rotmat_roundtrip_and_scipy():
    g = torch.Generator().manual_seed(0)
    aa = torch.randn(500, 3, generator=g, dtype=torch.float64)
    aa = aa / aa.norm(dim=1, keepdim=True) * (torch.rand(500, 1, generator=g, dtype=torch.float64) * 3.0 + 0.01)
    R = rot.tgm_angle_axis_to_rotation_matrix(aa)
    Rs = Rotation.from_rotvec(aa.numpy()).as_matrix()
    assert np.abs(R.numpy() - Rs).max() < 2e-6   # tgm divides by (theta + 1e-6): a ~1e-6 deviation by design
    back = rot.tgm_rotation_matrix_to_angle_axis(torch.from_numpy(Rs))
    assert np.abs(back.numpy() - aa.numpy()).max() < 1e-6
    # all four quaternion branches are reachable and consistent
    for axis in np.eye(3):
        Rpi = Rotation.from_rotvec(axis * 3.1).as_matrix()[None]
        v = rot.tgm_rotation_matrix_to_angle_axis(torch.from_numpy(Rpi))
        assert np.abs(Rotation.from_rotvec(v.numpy()).as_matrix() - Rpi).max() < 1e-9


def test_p3d_restatements_agree_with_scipy():
    g = torch.Generator().manual_seed(1)
    aa = torch.randn(200, 3, generator=g, dtype=torch.float64)
    M = rot.p3d_axis_angle_to_matrix(aa)
    assert np.abs(M.numpy() - Rotation.from_rotvec(aa.numpy()).as_matrix()).max() < 1e-12
    back = rot.p3d_matrix_to_axis_angle(M)
    assert np.abs(Rotation.from_rotvec(back.numpy()).as_matrix() - M.numpy()).max() < 1e-12


def test_smplx_oracle_invariants():
    bm = synth.make_body_model(0, num_verts=1500)
    ob32, ob64 = BodyModel(bm), BodyModel(bm, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    xb = torch.randn(6, 93, generator=g) * 0.3
    betas = torch.randn(6, 10, generator=g)
    v32, j32 = smplx_forward(ob32, xb, betas)
    v64, j64 = smplx_forward(ob64, xb.double(), betas.double())
    assert (v32.double() - v64).abs().max() < 5e-6 and (j32.double() - j64).abs().max() < 5e-6
    assert j32.shape == (6, 127, 3) and v32.shape == (6, 1500, 3)
    # translation equivariance
    xb2 = xb.double().clone()
    xb2[:, :3] += torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    v2, j2 = smplx_forward(ob64, xb2, betas.double())
    assert (v2 - v64 - torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)).abs().max() < 1e-12
    # zero pose and zero hand means -> v_shaped + transl
    bm0 = dict(bm)
    bm0["hand_mean_l"] = np.zeros(45, np.float32)
    bm0["hand_mean_r"] = np.zeros(45, np.float32)
    o0 = BodyModel(bm0, dtype=torch.float64)
    xz = torch.zeros(2, 93, dtype=torch.float64)
    xz[:, :3] = torch.tensor([0.1, 0.2, 0.3])
    vz, _, inter = smplx_forward(o0, xz, betas[:2].double(), return_intermediate=True)
    assert (vz - inter["v_shaped"] - xz[:, None, :3]).abs().max() < 1e-7
    # dense weights == the 4-nnz sparse structure the synthetic model was built with
    assert int((torch.as_tensor(bm["lbs_weights"]) != 0).sum(1).max()) == 4
