"""TEST INFRASTRUCTURE - CPU restatement of exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py on top of
oracle/smplx_lbs.py.  Pinned by tests/golden/canonicalize_ref.npz, which `scripts/gen_goldens.py canonicalize` produced by
running the reference's own `canonicalize_subsequence` (its smplx.create replaced by an adapter around the same oracle LBS:
smplx is absent from the reference tree).  Not imported by the product."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from .smplx_lbs import BodyModel, smplx_forward


def _bm(bm: BodyModel, transl, glorot, body_pose, betas):
    n = body_pose.shape[0]
    xb = torch.zeros(n, 93, dtype=bm.dtype)
    xb[:, :3] = torch.as_tensor(transl, dtype=bm.dtype)
    xb[:, 3:6] = torch.as_tensor(glorot, dtype=bm.dtype)
    xb[:, 6:69] = torch.as_tensor(body_pose, dtype=bm.dtype)
    b = torch.as_tensor(betas, dtype=bm.dtype).reshape(1, 10).expand(n, 10)
    v, j = smplx_forward(bm, xb, b)
    return v.float().numpy(), j.float().numpy()           # the reference's smplx runs in float32


def get_new_coordinate(bm, betas, transl, pose):
    """:60-86 - one frame: x = left hip -> right hip on the floor, z up, y = z x x; origin at the pelvis."""
    _, joints = _bm(bm, transl, pose[:, :3], pose[:, 3:], betas)
    joints = joints[0]
    x_axis = joints[2, :] - joints[1, :]
    x_axis[-1] = 0
    x_axis = x_axis / np.linalg.norm(x_axis)
    z_axis = np.array([0, 0, 1])
    y_axis = np.cross(z_axis, x_axis)
    y_axis = y_axis / np.linalg.norm(y_axis)
    return np.stack([x_axis, y_axis, z_axis], axis=1), joints[:1, :]


def calc_calibrate_offset(bm, betas, transl, pose):
    """:28-55 - pelvis of every frame's body at zero global orientation and translation."""
    n = transl.shape[0]
    _, joints = _bm(bm, np.zeros((n, 3)), np.zeros((n, 3)), pose[:, 3:], betas)
    return joints[:, 0, :]


def canonicalize_frames(bm, transl, pose, betas, cmu_ids, ssm_ids, fps=120.0):
    """:240-283 for one sub-sequence that is already cut and down-sampled."""
    transl, pose = np.array(transl, np.float64), np.array(pose, np.float64)
    b10 = np.asarray(betas)[:10]
    transf_rotmat, transf_transl = get_new_coordinate(bm, b10, transl[:1, :], pose[:1, :66])
    delta_T = calc_calibrate_offset(bm, b10, transl, pose[:, :66])
    global_ori = R.from_rotvec(pose[:, :3]).as_matrix()
    global_ori_new = np.einsum("ij,tjk->tik", transf_rotmat.T, global_ori)
    pose[:, :3] = R.from_matrix(global_ori_new).as_rotvec()
    transl = np.einsum("ij,tj->ti", transf_rotmat.T, transl + delta_T - transf_transl) - delta_T
    verts, joints = _bm(bm, transl, pose[:, :3], pose[:, 3:66], b10)
    return {"transf_rotmat": transf_rotmat, "transf_transl": transf_transl, "trans": transl, "poses": pose, "betas": b10,
            "gender": "male", "mocap_framerate": int(fps), "joints": joints[:, :22, :], "marker_cmu_41": verts[:, cmu_ids, :],
            "marker_ssm2_67": verts[:, ssm_ids, :]}
