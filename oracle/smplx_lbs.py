"""SMPL-X forward (blend shapes + Rodrigues chain + linear blend skinning), CPU restatement.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference calls `bm(return_verts=True, **bparam)` at models/baseops.py:382 (and :529) on a
`smplx.create(..., model_type='smplx', num_pca_comps=12, ...)` model (baseops.py:291-320).  The
arithmetic lives in the pip package `smplx` (pin evidence smplx==0.1.28 in
experiments/HOOD/hood.yml:230, experiments/HMR/prohmr.yml:176), which is NOT under
/root/reference: PARITY UNPINNED.  This file restates the published algorithm of
smplx/body_models.py::SMPLX.forward and smplx/lbs.py::{lbs, batch_rigid_transform,
vertices2landmarks}; the skinning tail agrees with the in-tree restatement
experiments/HOOD/utils/lbs.py:85-124.

Call-site contract reproduced (baseops.py:366-374): xb[B,93] = transl 3 | global_orient 3 |
body_pose 63 | left_hand PCA 12 | right_hand PCA 12; betas tiled to B; expression, jaw and eye
poses are zero.
"""
from typing import Dict, Tuple

import numpy as np
import torch

from .rot import smplx_batch_rodrigues


class BodyModel:
    """Holds the model tensors (from egogen_amd.synth.make_body_model or a real npz)."""

    def __init__(self, bm: Dict[str, np.ndarray], dtype=torch.float32):
        self.dtype = dtype
        t = lambda k: torch.as_tensor(np.asarray(bm[k]), dtype=dtype)
        self.v_template = t("v_template")            # [V,3]
        self.shapedirs = t("shapedirs")              # [V,3,10]
        self.posedirs = t("posedirs")                # [486,3V]
        self.J_regressor = t("J_regressor")          # [55,V]
        self.lbs_weights = t("lbs_weights")          # [V,55]
        self.parents = [int(x) for x in bm["parents"]]
        self.hand_comps_l = t("hand_comps_l")        # [12,45]
        self.hand_comps_r = t("hand_comps_r")
        self.hand_mean_l = t("hand_mean_l")          # [45]
        self.hand_mean_r = t("hand_mean_r")
        self.extra_vids = torch.as_tensor(np.asarray(bm["extra_vids"]), dtype=torch.long)   # [21]
        self.lmk_vids = torch.as_tensor(np.asarray(bm["lmk_vids"]), dtype=torch.long)       # [51,3]
        self.lmk_bary = t("lmk_bary")                # [51,3]
        self.V = self.v_template.shape[0]


def batch_rigid_transform(rot_mats, joints, parents) -> Tuple[torch.Tensor, torch.Tensor]:
    """smplx lbs.py batch_rigid_transform: returns posed joints [B,J,3] and relative
    transforms A [B,J,4,4] (rest joint subtracted)."""
    B, J = joints.shape[:2]
    dtype = joints.dtype
    rel = joints.clone()
    rel[:, 1:] = rel[:, 1:] - joints[:, parents[1:]]
    T = torch.zeros(B, J, 4, 4, dtype=dtype)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[parents[i]], T[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros(B, J, 1, dtype=dtype)], dim=2).unsqueeze(-1)  # [B,J,4,1]
    init_bone = torch.matmul(transforms, jh)  # [B,J,4,1]
    rel_transforms = transforms.clone()
    rel_transforms[:, :, :, 3:4] = rel_transforms[:, :, :, 3:4] - init_bone
    return posed_joints, rel_transforms


def smplx_forward(bm: BodyModel, xb: torch.Tensor, betas: torch.Tensor, return_intermediate=False):
    """xb[B,93], betas[B,10] (already tiled) -> vertices[B,V,3], joints[B,127,3]."""
    dt = bm.dtype
    xb = xb.to(dt)
    betas = betas.to(dt)
    B = xb.shape[0]
    transl, glorot, body_pose = xb[:, :3], xb[:, 3:6], xb[:, 6:69]
    lh = torch.einsum("bi,ij->bj", xb[:, 69:81], bm.hand_comps_l)
    rh = torch.einsum("bi,ij->bj", xb[:, 81:93], bm.hand_comps_r)
    zeros9 = torch.zeros(B, 9, dtype=dt)  # jaw, leye, reye
    full_pose = torch.cat([glorot, body_pose, zeros9, lh, rh], dim=1)  # [B,165]
    pose_mean = torch.cat([torch.zeros(3 + 63 + 9, dtype=dt), bm.hand_mean_l, bm.hand_mean_r])
    full_pose = full_pose + pose_mean
    # expression coefficients are zero on this path, so only the 10 beta directions contribute
    v_shaped = bm.v_template.unsqueeze(0) + torch.einsum("bl,mkl->bmk", betas, bm.shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, bm.J_regressor)
    rot_mats = smplx_batch_rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
    ident = torch.eye(3, dtype=dt)
    pose_feature = (rot_mats[:, 1:] - ident).reshape(B, -1)
    pose_offsets = torch.matmul(pose_feature, bm.posedirs).view(B, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, bm.parents)
    T = torch.matmul(bm.lbs_weights.unsqueeze(0).expand(B, -1, -1), A.reshape(B, -1, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, bm.V, 1, dtype=dt)], dim=2)
    verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]
    # landmarks (51 static, barycentric) + 21 vertex joints
    lmk = (verts[:, bm.lmk_vids] * bm.lmk_bary[None, :, :, None]).sum(dim=2)  # [B,51,3]
    joints = torch.cat([J_transformed, verts[:, bm.extra_vids], lmk], dim=1)
    joints = joints + transl.unsqueeze(1)
    verts = verts + transl.unsqueeze(1)
    if return_intermediate:
        return verts, joints, {"J": J, "A": A, "pose_feature": pose_feature, "v_posed": v_posed, "v_shaped": v_shaped}
    return verts, joints
