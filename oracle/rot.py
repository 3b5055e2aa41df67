"""Rotation conversions restated from the third-party packages the reference calls.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All functions are torch, dtype-generic.

* tgm_*  : torchgeometry 0.1.2 (pin evidence: experiments/HMR/prohmr.yml:186), called at
           models/baseops.py:139,161,171,587-590.  The in-tree kornia-derived copy
           experiments/HMR/prohmr/utils/konia_transform.py:234-313 has the same aa->R formula
           (theta^2 > 1e-6 Taylor switch).  PARITY UNPINNED for R->aa.
* smplx_batch_rodrigues : smplx 0.1.28 lbs.py batch_rodrigues (pin: experiments/HOOD/hood.yml:230).
* p3d_*  : pytorch3d 0.7.4 transforms (pin: experiments/HOOD/hood.yml:167), called at
           exp_GAMMAPrimitive/utils/environments.py:167,233,237.  PARITY UNPINNED.
"""
import torch


def tgm_angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,3,3] (upper-left block of tgm's 4x4)."""
    aa = angle_axis
    theta2 = (aa * aa).sum(dim=1, keepdim=True)  # [N,1]
    theta = torch.sqrt(theta2)
    w = aa / (theta + 1e-6)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c = torch.cos(theta)
    s = torch.sin(theta)
    r00 = c + wx * wx * (1 - c)
    r10 = wz * s + wx * wy * (1 - c)
    r20 = -wy * s + wx * wz * (1 - c)
    r01 = wx * wy * (1 - c) - wz * s
    r11 = c + wy * wy * (1 - c)
    r21 = wx * s + wy * wz * (1 - c)
    r02 = wy * s + wx * wz * (1 - c)
    r12 = -wx * s + wy * wz * (1 - c)
    r22 = c + wz * wz * (1 - c)
    normal = torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    mask = (theta2 > 1e-6).view(-1, 1, 1).to(aa.dtype)
    return mask * normal + (1 - mask) * taylor


def tgm_rotation_matrix_to_quaternion(R: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """[N,3,3] -> [N,4] (w,x,y,z), torchgeometry 0.1.2 4-branch form (works on R^T)."""
    m = R.transpose(1, 2)
    mask_d2 = m[:, 2, 2] < eps
    mask_d0_d1 = m[:, 0, 0] > m[:, 1, 1]
    mask_d0_nd1 = m[:, 0, 0] < -m[:, 1, 1]
    t0 = 1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2]
    q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
    t1 = 1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2]
    q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
    t2 = 1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2]
    q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
    t3 = 1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
    c0 = (mask_d2 & mask_d0_d1).to(R.dtype).view(-1, 1)
    c1 = (mask_d2 & ~mask_d0_d1).to(R.dtype).view(-1, 1)
    c2 = (~mask_d2 & mask_d0_nd1).to(R.dtype).view(-1, 1)
    c3 = (~mask_d2 & ~mask_d0_nd1).to(R.dtype).view(-1, 1)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.view(-1, 1) * c0 + t1.view(-1, 1) * c1 + t2.view(-1, 1) * c2 + t3.view(-1, 1) * c3)
    return q * 0.5


def tgm_quaternion_to_angle_axis(q: torch.Tensor) -> torch.Tensor:
    q1, q2, q3 = q[..., 1], q[..., 2], q[..., 3]
    sin_sq = q1 * q1 + q2 * q2 + q3 * q3
    sin_t = torch.sqrt(sin_sq)
    cos_t = q[..., 0]
    two_theta = 2.0 * torch.where(cos_t < 0.0, torch.atan2(-sin_t, -cos_t), torch.atan2(sin_t, cos_t))
    k_pos = two_theta / sin_t
    k_neg = 2.0 * torch.ones_like(sin_t)
    k = torch.where(sin_sq > 0.0, k_pos, k_neg)
    return torch.stack([q1 * k, q2 * k, q3 * k], -1)


def tgm_rotation_matrix_to_angle_axis(R: torch.Tensor) -> torch.Tensor:
    """[N,3,3] -> [N,3]  (baseops.py:160-161: F.pad to 3x4 then tgm; the pad column is unused)."""
    return tgm_quaternion_to_angle_axis(tgm_rotation_matrix_to_quaternion(R))


def smplx_batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,3,3]; angle = ||r + 1e-8|| (smplx lbs.py batch_rodrigues)."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle).unsqueeze(1)
    sin = torch.sin(angle).unsqueeze(1)
    rx, ry, rz = rot_dir[:, 0:1], rot_dir[:, 1:2], rot_dir[:, 2:3]
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def cont2rotmat(x6: torch.Tensor) -> torch.Tensor:
    """models/baseops.py:119-130 RotConverter.cont2rotmat: [...,6] viewed as (3,2) -> [N,3,3]."""
    a = x6.contiguous().view(-1, 3, 2)
    b1 = torch.nn.functional.normalize(a[:, :, 0], dim=1)
    dot = torch.sum(b1 * a[:, :, 1], dim=1, keepdim=True)
    b2 = torch.nn.functional.normalize(a[:, :, 1] - dot * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def cont2aa(x6: torch.Tensor) -> torch.Tensor:
    """models/baseops.py:143-162 cont2aa -> rotmat2aa (tgm)."""
    return tgm_rotation_matrix_to_angle_axis(cont2rotmat(x6))


# ---- pytorch3d 0.7.4 (reset path only) -------------------------------------------------------

def p3d_axis_angle_to_matrix(aa: torch.Tensor) -> torch.Tensor:
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s_over = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), aa * s_over], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def p3d_matrix_to_axis_angle(M: torch.Tensor) -> torch.Tensor:
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(M.reshape(M.shape[:-2] + (9,)), dim=-1)
    x = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1)
    q_abs = torch.sqrt(torch.clamp(x, min=0.0))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    idx = q_abs.argmax(dim=-1)
    q = torch.gather(cand, -2, idx[..., None, None].expand(idx.shape + (1, 4))).squeeze(-2)
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s_over = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    return q[..., 1:] / s_over


def rotz(theta: torch.Tensor) -> torch.Tensor:
    """pytorch3d euler_angles_to_matrix([0,0,theta], 'XYZ') == rotation about z."""
    c, s = torch.cos(theta), torch.sin(theta)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack([c, -s, z, s, c, z, z, z, o], -1).reshape(theta.shape + (3, 3))
