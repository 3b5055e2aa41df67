"""Training of the body regressor: host-side mirror of `GAMMARegressorTrainOP`
(models/models_GAMMA_primitive.py:594-711) and of the batch-generator methods it is fed by
(`next_batch_genderselection`, `next_sequence`: exp_GAMMAPrimitive/utils/batch_gen_amass.py:271-426), SURVEY 8(f) N1.

Same constructor arguments, loss (`calc_loss`: L1 between the reference markers and the SSM2-67 markers of the SMPL-X body
posed by the regressed parameters, plus `weight_reg_hpose` x mean square of the 24 hand PCA coefficients), optimiser /
scheduler and checkpoint layout (`<save_dir>/epoch-N.ckp`, the file GAMMAPrimitiveComboGenOP.build_model loads at :1141).

The network (66 dense layers, three recurrences over shared weights) runs through the HIP-backed autograd nodes of
`egogen_amd.fused_ops`.  The loss only needs the 67 marker vertices, so the body model is evaluated on those rows alone
(`MarkerBodyModel`, torch ops on the device: [B,67,3] instead of [B,10475,3], differentiable in the pose); it is a training-time
operator, the rollout keeps using the fused LBS kernel.  There is no CPU fallback.
"""
from __future__ import annotations

import glob
import logging
import os
import time
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .fused_ops import FlatGrads, linear_act, res_mlp
from .models import MoshRegressor
from .train_predictor import BatchGeneratorAMASSCanonicalized as _PredictorBatcher, get_scheduler


# ---------------------------------------------------------------------------------------------------------------------
# rotation tail of MoshRegressor.forward with use_cont (models_GAMMA_primitive.py:208-219 -> baseops.py:119-162)
# ---------------------------------------------------------------------------------------------------------------------
def cont6d_to_rotmat(x6: torch.Tensor) -> torch.Tensor:
    """RotConverter.cont2rotmat (baseops.py:119-130): [n,6] read as a (3,2) matrix -> Gram-Schmidt -> [n,3,3] (columns b1,b2,b3)."""
    a = x6.reshape(-1, 3, 2)
    b1 = F.normalize(a[..., 0], dim=1)
    b2 = F.normalize(a[..., 1] - (b1 * a[..., 1]).sum(1, keepdim=True) * b1, dim=1)
    return torch.stack([b1, b2, torch.linalg.cross(b1, b2, dim=1)], dim=-1)


def rotmat_to_aa(R: torch.Tensor) -> torch.Tensor:
    """RotConverter.rotmat2aa (baseops.py:152-162): torchgeometry's rotation_matrix_to_angle_axis [upstream 0.1.2: the
    four-branch quaternion extraction on R^T, then 2 atan2(|v|, w) / |v|], written with selects so that autograd follows the
    branch each row takes.  The same arithmetic as the device function egx_tgm_rotmat_to_aa used by the rollout kernels."""
    m00, m01, m02 = R[:, 0, 0], R[:, 1, 0], R[:, 2, 0]   # m = R^T
    m10, m11, m12 = R[:, 0, 1], R[:, 1, 1], R[:, 2, 1]
    m20, m21, m22 = R[:, 0, 2], R[:, 1, 2], R[:, 2, 2]
    neg_z = m22 < 1e-6
    x_big, xy_neg = m00 > m11, m00 < -m11
    cand = (  # (t, w, x, y, z) of each branch
        (1 + m00 - m11 - m22, m12 - m21, None, m01 + m10, m20 + m02),
        (1 - m00 + m11 - m22, m20 - m02, m01 + m10, None, m12 + m21),
        (1 - m00 - m11 + m22, m01 - m10, m20 + m02, m12 + m21, None),
        (1 + m00 + m11 + m22, None, m12 - m21, m20 - m02, m01 - m10),
    )
    rows = [torch.stack([t if c is None else c for c in (w, x, y, z)], -1) for (t, w, x, y, z) in cand]
    ts = [c[0] for c in cand]
    sel = [neg_z & x_big, neg_z & ~x_big, ~neg_z & xy_neg, ~neg_z & ~xy_neg]
    q, t = rows[3], ts[3]
    for k in range(3):
        q = torch.where(sel[k].unsqueeze(-1), rows[k], q)
        t = torch.where(sel[k], ts[k], t)
    q = 0.5 * q / torch.sqrt(t).unsqueeze(-1)
    v = q[:, 1:]
    s2 = (v * v).sum(-1)
    s = torch.sqrt(s2)
    w = q[:, 0]
    two_theta = 2.0 * torch.where(w < 0, torch.atan2(-s, -w), torch.atan2(s, w))
    safe = torch.where(s2 > 0, s, torch.ones_like(s))
    k = torch.where(s2 > 0, two_theta / safe, torch.full_like(s, 2.0))
    return v * k.unsqueeze(-1)


def cont6d_params_to_aa(xb6: torch.Tensor) -> torch.Tensor:
    """MoshRegressor._cont2aa (:208-219): [n,159] = transl 3 | 22 x 6D | hands 24  ->  [n,93] = transl | 22 x axis-angle | hands."""
    n = xb6.shape[0]
    aa = rotmat_to_aa(cont6d_to_rotmat(xb6[:, 3:135].reshape(-1, 6))).reshape(n, 66)
    return torch.cat([xb6[:, :3], aa, xb6[:, 135:]], dim=-1)


# ---------------------------------------------------------------------------------------------------------------------
# SMPL-X on the marker rows only
# ---------------------------------------------------------------------------------------------------------------------
class _ConstMatmulFn(torch.autograd.Function):
    """y = x P for a constant P[K,N] (a model buffer): forward and the input gradient g P^T are `egx_gemm3` products."""

    @staticmethod
    def forward(ctx, x, P):
        from .fused_ops import gemm3
        ctx.save_for_backward(P)
        return gemm3(x.contiguous(), False, P, True)

    @staticmethod
    def backward(ctx, g):
        from .fused_ops import gemm3
        (P,) = ctx.saved_tensors
        return (gemm3(g.contiguous(), False, P, False) if ctx.needs_input_grad[0] else None), None


def _cmm(x, P):
    if x.is_cuda and x.dtype == torch.float32:
        return _ConstMatmulFn.apply(x, P)
    return x @ P          # host / float64 evaluation (tests compare the two)


def _mm3(a, b):
    """[...,3,3] x [...,3,3] written out (9 dot products of length 3: element-wise kernels, not a batched GEMM)."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


def _mv3(a, v):
    return (a * v.unsqueeze(-2)).sum(-1)


class MarkerBodyModel(torch.nn.Module):
    """`bm(return_verts=True, **body_param).vertices[:, markers]` (models_GAMMA_primitive.py:629) without the other 10 408
    vertices.  Arithmetic of smplx.SMPLX.forward / lbs.lbs [upstream smplx 0.1.28] restricted to the rows in `marker_vids`:
    the rest joints are an affine function of betas (J_regressor folded into the template and the shape directions once, in
    float64), the pose chain is the full 55-joint one, blend shapes and skinning weights are gathered at the markers.  The
    four matrix products (hand PCA, shape and pose blend shapes, skinning-weight blend of the joint transforms) have a constant
    right-hand side and run as `egx_gemm3` products; the 3x3 chain is element-wise."""

    def __init__(self, bm: Dict[str, np.ndarray], marker_vids):
        super().__init__()
        vids = np.asarray(marker_vids, np.int64)
        m = len(vids)
        V = bm["v_template"].shape[0]
        Jr = np.asarray(bm["J_regressor"], np.float64)
        buf = lambda name, a: self.register_buffer(name, torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32), persistent=False)
        buf("J_template", Jr @ np.asarray(bm["v_template"], np.float64))                                              # [55,3]
        buf("J_shapedirs", np.einsum("jv,vkl->jkl", Jr, np.asarray(bm["shapedirs"], np.float64)).reshape(-1, 10).T)  # [10,165]
        buf("v_template", np.asarray(bm["v_template"])[vids])                                                          # [m,3]
        buf("shapedirs", np.asarray(bm["shapedirs"])[vids].reshape(3 * m, 10).T)                                       # [10,3m]
        buf("posedirs", np.asarray(bm["posedirs"]).reshape(-1, V, 3)[:, vids].reshape(-1, 3 * m))                     # [486,3m]
        buf("lbs_weights_t", np.asarray(bm["lbs_weights"])[vids].T)                                                    # [55,m]
        hand = np.zeros((24, 90), np.float64)                                                                           # block diagonal
        hand[:12, :45], hand[12:, 45:] = np.asarray(bm["hand_comps_l"]), np.asarray(bm["hand_comps_r"])
        buf("hand_comps", hand)
        buf("hand_mean", np.concatenate([np.asarray(bm["hand_mean_l"]), np.asarray(bm["hand_mean_r"])]))
        parents = [int(p) for p in bm["parents"]]
        self.parents = parents
        # joints grouped by depth so that the chain is one batched step per level, not one per joint
        depth = [0] * len(parents)
        for j in range(1, len(parents)):
            depth[j] = depth[parents[j]] + 1
        self.levels = [[j for j in range(1, len(parents)) if depth[j] == d] for d in range(1, max(depth) + 1)]

    @staticmethod
    def rodrigues(r: torch.Tensor) -> torch.Tensor:
        """smplx lbs.batch_rodrigues: angle = |r + 1e-8|, d = r / angle, R = I + sin K + (1 - cos) K^2 with K the cross-product
        matrix of d; K^2 = d d^T - |d|^2 I for any d, so no matrix product is needed."""
        angle = torch.linalg.vector_norm(r + 1e-8, dim=1, keepdim=True)
        d = r / angle
        zero = torch.zeros_like(angle[:, 0])
        K = torch.stack([zero, -d[:, 2], d[:, 1], d[:, 2], zero, -d[:, 0], -d[:, 1], d[:, 0], zero], 1).view(-1, 3, 3)
        eye = torch.eye(3, dtype=r.dtype, device=r.device)
        K2 = d.unsqueeze(-1) * d.unsqueeze(-2) - (d * d).sum(-1).view(-1, 1, 1) * eye
        s, c = torch.sin(angle).unsqueeze(-1), torch.cos(angle).unsqueeze(-1)
        return eye + s * K + (1 - c) * K2

    def forward(self, xb: torch.Tensor, betas: torch.Tensor) -> torch.Tensor:
        """xb[B,93] (transl | global_orient | body_pose 63 | hand PCA 12 + 12), betas[B,10] -> markers[B,m,3]."""
        B = xb.shape[0]
        hands = _cmm(xb[:, 69:93], self.hand_comps) + self.hand_mean
        full_pose = torch.cat([xb[:, 3:69], xb.new_zeros(B, 9), hands], 1)                 # jaw and eyes stay at rest
        rot = self.rodrigues(full_pose.reshape(-1, 3)).view(B, 55, 3, 3)
        J = self.J_template + _cmm(betas, self.J_shapedirs).view(B, 55, 3)
        v_shaped = self.v_template + _cmm(betas, self.shapedirs).view(B, -1, 3)
        pose_feature = (rot[:, 1:] - torch.eye(3, dtype=xb.dtype, device=xb.device)).reshape(B, -1)
        v_posed = v_shaped + _cmm(pose_feature, self.posedirs).view(B, -1, 3)
        # world transform of every joint: G_j = G_parent(j) [R_j | J_j - J_parent(j)]
        rel = torch.cat([J[:, :1], J[:, 1:] - J[:, self.parents[1:]]], 1)
        G_R, G_t = [None] * 55, [None] * 55
        G_R[0], G_t[0] = rot[:, 0], rel[:, 0]
        for lv in self.levels:
            pr = torch.stack([G_R[self.parents[j]] for j in lv], 1)                          # [B,n,3,3]
            pt = torch.stack([G_t[self.parents[j]] for j in lv], 1)
            r = _mm3(pr, rot[:, lv])
            t = _mv3(pr, rel[:, lv]) + pt
            for i, j in enumerate(lv):
                G_R[j], G_t[j] = r[:, i], t[:, i]
        GR, Gt = torch.stack(G_R, 1), torch.stack(G_t, 1)                                    # [B,55,3,3], [B,55,3]
        At = Gt - _mv3(GR, J)                                                                # rest joint removed
        # skinning transforms of the markers: T_m = sum_j w[m,j] [GR_j | At_j], one product with the 12 entries as rows
        X = torch.cat([GR.reshape(B, 55, 9), At], -1).permute(0, 2, 1).reshape(B * 12, 55)
        T = _cmm(X, self.lbs_weights_t).view(B, 12, -1).permute(0, 2, 1)                     # [B,m,12]
        TR, Tt = T[..., :9].reshape(B, -1, 3, 3), T[..., 9:]
        return _mv3(TR, v_posed) + Tt + xb[:, None, :3]


# ---------------------------------------------------------------------------------------------------------------------
# the train operator
# ---------------------------------------------------------------------------------------------------------------------
class MoshRegressorTrain(MoshRegressor):
    """MoshRegressor.forward (:262-301) for training: the same module / state_dict as the rollout's, evaluated through autograd
    nodes whose products run as device GEMMs and whose weight gradients accumulate into a flat buffer."""

    def forward(self, marker_ref: torch.Tensor, betas: torch.Tensor) -> torch.Tensor:
        n = marker_ref.shape[0]
        xr = marker_ref.reshape(n, self.in_dim)
        xb = xr.new_zeros(n, self.body_dim)
        net = self.pnet
        for _ in range(3):
            h = linear_act(torch.cat([xr, xb, betas], dim=-1), net.in_fc)
            for blk in net.layers:
                h = res_mlp(h, blk.layers[0], blk.layers[1], act="relu")
            if (n * self.body_dim) % 4 == 0:      # the fused residual pass works on float4s
                xb = linear_act(h, net.out_fc, res=xb)
            else:
                xb = linear_act(h, net.out_fc) + xb
        return cont6d_params_to_aa(xb)


class GAMMARegressorTrainOP:
    def __init__(self, modelconfig, lossconfig, trainconfig):
        if not torch.cuda.is_available():
            raise _lib.EgxError("regressor training needs a HIP device (no CPU fallback)")
        self.dtype = torch.float32
        self.device = torch.device("cuda", index=trainconfig.get("gpu_index", 0))
        self.modelconfig, self.lossconfig, self.trainconfig = modelconfig, lossconfig, trainconfig
        os.makedirs(trainconfig["log_dir"], exist_ok=True)
        self.logger = logging.getLogger(trainconfig["log_dir"])
        if not self.logger.handlers:
            self.logger.addHandler(logging.FileHandler(os.path.join(trainconfig["log_dir"], "train.log")))
        self.logger.setLevel(logging.INFO)

    def build_model(self, body_model: Optional[Dict[str, np.ndarray]] = None, markers=None):
        """:595-613.  `body_model` / `markers`: the SMPL-X tensors and SSM2 vertex ids; by default the ones the rest of the
        package uses (setup_world.load_body_model: the real SMPLX_<GENDER>.npz when present, else the synthetic body)."""
        if self.modelconfig["body_repr"] != "ssm2_67":
            raise ValueError("other marker placement is not considered yet.")
        self.model = MoshRegressorTrain(self.modelconfig).to(self.device)
        self.model.train()
        self.grads = FlatGrads(self.model)
        self.use_cont = self.modelconfig.get("use_cont", False)
        from . import synth
        if body_model is None:
            from . import setup_world
            body_model, _ = setup_world.load_body_model(self.modelconfig["gender"])
        if markers is None:
            markers = synth.marker_ids(body_model["v_template"].shape[0])
        self.markers = self.model.markers = [int(v) for v in markers]
        self.bm = MarkerBodyModel(body_model, self.markers).to(self.device)

    def calc_loss(self, x_ref, xb, betas):
        """:617-633: x_ref[n,67,3], xb[n,93] (axis-angle rotations), betas[n,10]."""
        x_pred = self.bm(xb, betas)
        loss_marker = F.l1_loss(x_ref, x_pred)
        loss_hpose = torch.mean(xb[:, 69:] ** 2)
        loss = loss_marker + self.lossconfig["weight_reg_hpose"] * loss_hpose
        return loss, torch.stack([loss_marker.detach(), loss_hpose.detach()]).cpu().numpy()

    def step(self, optimizer, marker_ref, batch_betas):
        """One optimiser step of the loop body (:670-680) on marker_ref[n,67,3], batch_betas[n,10]."""
        xb_new = self.model(marker_ref.detach(), batch_betas)
        self.grads.zero()
        loss, items = self.calc_loss(marker_ref, xb_new, batch_betas)
        loss.backward(retain_graph=False)
        optimizer.step()
        return loss.detach(), items

    def train(self, batch_gen):
        """:636-709"""
        if getattr(self, "model", None) is None:
            self.build_model()
        tc = self.trainconfig
        batch_size, gender = tc["batch_size"], self.modelconfig["gender"]
        starting_epoch = 0
        optimizer = torch.optim.Adam(self.model.parameters(), lr=tc["learning_rate"])
        scheduler = get_scheduler(optimizer, policy="lambda", num_epochs_fix=tc["num_epochs_fix"], num_epochs=tc["num_epochs"])
        if tc.get("resume_training", False):
            ckp_list = sorted(glob.glob(os.path.join(tc["save_dir"], "epoch-*.ckp")), key=os.path.getmtime)
            if ckp_list:
                checkpoint = torch.load(ckp_list[-1], map_location=self.device)
                self.model.load_state_dict(checkpoint["model_state_dict"])
                optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
                starting_epoch = checkpoint["epoch"]
                self.grads.attach()
        loss_names = ["MSE_MARKER", "MSE_HPOSE"]
        history = []
        for epoch in range(starting_epoch, tc["num_epochs"]):
            epoch_losses, epoch_nsamples = 0, 0
            stime = time.time()
            while batch_gen.has_next_rec():
                data = batch_gen.next_batch_genderselection(batch_size, gender)
                if data is None:
                    continue
                batch_betas, marker_ref = data[:2]
                marker_ref = marker_ref.contiguous().view(-1, self.model.in_dim)
                marker_ref = marker_ref.view(marker_ref.shape[0], -1, 3).to(self.device)
                batch_betas = batch_betas.contiguous().view(-1, 10).to(self.device)
                _, items = self.step(optimizer, marker_ref, batch_betas)
                epoch_losses = epoch_losses + items
                epoch_nsamples += 1
            batch_gen.reset()
            scheduler.step()
            epoch_losses = epoch_losses / max(epoch_nsamples, 1)
            info = "[epoch {:d}]:".format(epoch + 1) + "".join("{}={:f}, ".format(n, v) for n, v in zip(loss_names, np.atleast_1d(epoch_losses)))
            info += "time={:f}, lr={:f}".format(time.time() - stime, optimizer.param_groups[0]["lr"])
            self.logger.info(info)
            history.append(np.atleast_1d(epoch_losses))
            if (1 + epoch) % tc["saving_per_X_ep"] == 0:
                os.makedirs(tc["save_dir"], exist_ok=True)
                torch.save({"epoch": epoch + 1, "model_state_dict": self.model.state_dict(),
                            "optimizer_state_dict": optimizer.state_dict()}, os.path.join(tc["save_dir"], f"epoch-{epoch + 1}.ckp"))
            if tc.get("verbose", False):
                print(info)
        return history


# ---------------------------------------------------------------------------------------------------------------------
# batch generator: the per-file readers of batch_gen_amass.py (the predictor's batcher keeps everything in RAM instead)
# ---------------------------------------------------------------------------------------------------------------------
BODY_REPRS = ("smpl_params", "joints", "cmu_41", "ssm2_67", "ssm2_67_marker2tarloc", "bone_transform")


def target_features(joints, body_ssm2_67, transl=np.zeros((1, 3))):
    """_get_target_feature (batch_gen_amass.py:271-283): joints[t,22,3], markers[t,67,3] -> (marker-to-last-frame offsets,
    unit walking direction [t,2], unit marker-to-target-pelvis directions [t,67,3]).  As upstream, the target pelvis height
    is lowered by `transl` IN PLACE in `joints` (a view of the last frame)."""
    wpath = (joints[-1:] - joints)[:, 0, :2]
    wpath_n = wpath / (1e-8 + np.linalg.norm(wpath, axis=-1, keepdims=True))
    vec_to_target = body_ssm2_67[-1:] - body_ssm2_67
    target_loc = joints[-1:, 0:1]
    target_loc[:, :, -1] = target_loc[:, :, -1] - transl[None, ...][:, :, -1]
    vec = target_loc - body_ssm2_67
    return vec_to_target, wpath_n, vec / np.linalg.norm(vec, axis=-1, keepdims=True)


def _body_feature(body_repr, transl, pose, joints, cmu_41, ssm2_67, marker2tarloc_n):
    if body_repr == "smpl_params":
        return np.concatenate([transl, pose], axis=-1)
    if body_repr == "joints":
        return joints.reshape([-1, 22 * 3])
    if body_repr == "cmu_41":
        return cmu_41.reshape([-1, 41 * 3])
    if body_repr == "ssm2_67":
        return ssm2_67.reshape([-1, 67 * 3])
    if body_repr == "ssm2_67_marker2tarloc":
        return np.concatenate([ssm2_67.reshape([-1, 67 * 3]), marker2tarloc_n.reshape([-1, 67 * 3])], axis=-1)
    if body_repr == "bone_transform":
        return np.concatenate([joints, pose.reshape([-1, 22, 3])], axis=-1)
    raise NameError("[ERROR] not valid body representation. Terminate")


class BatchGeneratorAMASSCanonicalized(_PredictorBatcher):
    """The file-at-a-time side of batch_gen_amass.py:61-430 next to the in-RAM batcher of train_predictor: the record list
    without loading (`get_rec_list(..., read_to_ram=False)` semantics), `next_sequence` (:287-343) and
    `next_batch_genderselection` (:348-426).  Every body representation of :312-329 is available here."""

    def __init__(self, amass_data_path, amass_subset_name=None, sample_rate=3, body_repr="ssm2_67", read_to_ram=False, device="cuda"):
        if body_repr not in BODY_REPRS:
            raise NameError("[ERROR] not valid body representation. Terminate")
        if read_to_ram:
            super().__init__(amass_data_path, amass_subset_name, sample_rate, body_repr, True, device)
        else:
            self.rec_list, self.index_rec = [], 0
            self.amass_data_path, self.amass_subset_name, self.sample_rate = amass_data_path, amass_subset_name, sample_rate
            self.body_repr, self.device = body_repr, device
            self.max_len = 200 if "x10" in amass_data_path else 20
            self.data_all = self.jts_all = None
        self.read_to_ram = read_to_ram

    def get_rec_list(self, shuffle_seed=None, to_gpu=True):
        if self.read_to_ram:
            return super().get_rec_list(shuffle_seed, to_gpu)
        import random
        if self.amass_subset_name is not None:
            self.rec_list = []
            for subset in self.amass_subset_name:
                self.rec_list += sorted(glob.glob(os.path.join(self.amass_data_path, subset, "*.npz")))
        else:
            self.rec_list = sorted(glob.glob(os.path.join(self.amass_data_path, "*/*.npz")))
        (random.Random(shuffle_seed) if shuffle_seed is not None else random).shuffle(self.rec_list)

    def has_next_rec(self):
        if self.read_to_ram:
            return super().has_next_rec()
        return self.index_rec < len(self.rec_list)

    def reset(self):
        if self.read_to_ram:
            return super().reset()
        import random
        self.index_rec = 0
        random.shuffle(self.rec_list)

    def _read(self, rec):
        with np.load(rec) as data:
            sr = self.sample_rate
            out = dict(transl=data["trans"][::sr], pose=data["poses"][::sr, :66], betas=data["betas"][:10], gender=data["gender"],
                       cmu_41=data["marker_cmu_41"][::sr], ssm2_67=data["marker_ssm2_67"][::sr],
                       joints=data["joints"][::sr].reshape([-1, 22, 3]), transf_rotmat=data["transf_rotmat"],
                       transf_transl=data["transf_transl"])
        return out

    def next_sequence(self):
        """:287-343: one record with its meta information; None for records with non-finite pose / translation (the index
        does not advance in that case, as upstream)."""
        d = self._read(self.rec_list[self.index_rec])
        if not (np.isfinite(d["pose"]).all() and np.isfinite(d["transl"]).all()):
            return None
        _, _, m2t = target_features(d["joints"], d["ssm2_67"])
        feature = _body_feature(self.body_repr, d["transl"], d["pose"], d["joints"], d["cmu_41"], d["ssm2_67"], m2t)
        self.index_rec += 1
        return {"betas": d["betas"], "gender": d["gender"], "transl": d["transl"], "glorot": d["pose"][:, :3], "poses": d["pose"][:, 3:],
                "body_feature": feature, "transf_rotmat": d["transf_rotmat"], "transf_transl": d["transf_transl"],
                "pelvis_loc": d["joints"][:, 0, :]}

    def next_batch_genderselection(self, batch_size=64, gender="male", batch_first=True, noise=None):
        """:348-426: the next `batch_size` records of one gender -> [betas, body_feature, transl, glorot, thetas, joints] as
        fp32 device tensors [b,t,d] (or [t,b,d]); None when the list runs out first."""
        cols = [[] for _ in range(6)]
        bb = 0
        while self.has_next_rec():
            rec = self.rec_list[self.index_rec]
            if bb == batch_size:
                break
            with np.load(rec) as data:
                rec_gender = str(data["gender"])
            if rec_gender != gender:
                self.index_rec += 1
                continue
            d = self._read(rec)
            _, _, m2t = target_features(d["joints"], d["ssm2_67"])
            feature = _body_feature(self.body_repr, d["transl"], d["pose"], d["joints"], d["cmu_41"], d["ssm2_67"], m2t)
            for c, v in zip(cols, (np.tile(d["betas"], (d["transl"].shape[0], 1)), feature, d["transl"], d["pose"][:, :3], d["pose"][:, 3:],
                                   d["joints"].reshape([-1, 22 * 3]))):
                c.append(v)
            self.index_rec += 1
            bb += 1
        if len(cols[0]) < batch_size:
            return None
        dim = 0 if batch_first else 1
        betas, feature, transl, glorot, thetas, jts = [torch.as_tensor(np.stack(c, axis=dim), dtype=torch.float32, device=self.device) for c in cols]
        return [betas, feature, transl, glorot, thetas, jts]
