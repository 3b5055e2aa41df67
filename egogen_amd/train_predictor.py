"""Training of the marker predictor (C-VAE): host-side mirror of `GAMMAPrimitiveVAETrainOP`
(models/models_GAMMA_primitive.py:389-591) and of the AMASS/SAMP batch generator it is fed by
(exp_GAMMAPrimitive/utils/batch_gen_amass.py:61-270), SURVEY 8(f) N1/N2.

Same constructor arguments, loss definitions (`_calc_loss_rec`, `calc_loss`, `calc_loss_rollout`), optimiser / scheduler and
checkpoint layout (`<save_dir>/epoch-N.ckp` = {'epoch', 'model_state_dict', 'optimizer_state_dict'}, the file
`GAMMAPrimitiveComboGenOP.build_model` loads, :1116-1148).  The forward / backward of the network runs through the HIP-backed
autograd nodes of `egogen_amd.fused_ops` (`GAMMAPrimitiveVAE.forward_train`); there is no CPU fallback.
"""
from __future__ import annotations

import glob
import logging
import os
import random
import time
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .fused_ops import FlatGrads
from .models import GAMMAPrimitiveVAE


def canonical_frame(joints: torch.Tensor):
    """CanonicalCoordinateExtractor.get_new_coordinate (baseops.py:214-225) on the device: joints[b,J,3] -> R[b,3,3], T[b,1,3]."""
    lib = _lib.load()
    j = joints.to(torch.float32).contiguous()
    b = j.shape[0]
    R = torch.empty(b, 3, 3, dtype=torch.float32, device=j.device)
    T = torch.empty(b, 1, 3, dtype=torch.float32, device=j.device)
    _lib.check(lib.egx_canonical_frame(_lib.ptr(j), int(j.shape[1]), b, _lib.ptr(R), _lib.ptr(T), _lib.current_stream_ptr()),
               "egx_canonical_frame")
    return R, T


def get_scheduler(optimizer, policy, num_epochs_fix=None, num_epochs=None):
    """baseops.py:52-61"""
    if policy != "lambda":
        raise NotImplementedError(f"scheduler with {policy} is not implemented")

    def lambda_rule(epoch):
        return 1.0 - max(0, epoch - num_epochs_fix) / float(num_epochs - num_epochs_fix + 1)
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)


class GAMMAPrimitiveVAETrainOP:
    def __init__(self, modelconfig, lossconfig, trainconfig):
        if not torch.cuda.is_available():
            raise _lib.EgxError("predictor training needs a HIP device (no CPU fallback)")
        self.dtype = torch.float32
        self.device = torch.device("cuda", index=trainconfig.get("gpu_index", 0))
        self.modelconfig, self.lossconfig, self.trainconfig = modelconfig, lossconfig, trainconfig
        os.makedirs(trainconfig["log_dir"], exist_ok=True)
        self.logger = logging.getLogger(trainconfig["log_dir"])
        if not self.logger.handlers:
            self.logger.addHandler(logging.FileHandler(os.path.join(trainconfig["log_dir"], "train.log")))
        self.logger.setLevel(logging.INFO)

    def build_model(self):
        self.model = GAMMAPrimitiveVAE(self.modelconfig).to(self.device)
        self.model.train()
        self.grads = FlatGrads(self.model)
        self.max_rollout = self.trainconfig.get("max_rollout", None)
        self.noise = self.trainconfig.get("noise", None)
        self.t_his = self.modelconfig["t_his"]

    # ---- losses (models_GAMMA_primitive.py:401-505) -------------------------------------------------------------
    def _calc_loss_rec(self, Y, Y_rec):
        loss_rec = torch.nn.functional.l1_loss(Y, Y_rec)
        loss_td = torch.nn.functional.l1_loss(Y_rec[1:] - Y_rec[:-1], Y[1:] - Y[:-1])
        return self.lossconfig["weight_rec"] * loss_rec + self.lossconfig["weight_td"] * loss_td

    def _kld(self, mu, logvar, epoch):
        loss_kld = 0.5 * torch.mean(-1 - logvar + mu.pow(2) + logvar.exp())
        if self.lossconfig["robust_kld"]:
            loss_kld = torch.sqrt(1 + loss_kld ** 2) - 1
        weight_kld = self.lossconfig["weight_kld"]
        if self.lossconfig["annealing_kld"]:
            weight_kld = min(float(epoch) / (0.9 * self.trainconfig["num_epochs"]), 1.0) * self.lossconfig["weight_kld"]
        return loss_kld, weight_kld

    def calc_loss(self, data, epoch, eps: Optional[torch.Tensor] = None):
        """data[t,b,201] -> loss, np.array([loss, rec, kld]).  `eps`: the reparameterisation noise (tests inject it)."""
        t_his = self.t_his
        X = data[:t_his]
        Y = data[t_his:, :, :self.model.in_dim]
        Y_rec, mu, logvar = self.model.forward_train(X, Y, eps)
        loss_rec = self._calc_loss_rec(Y, Y_rec)
        loss_kld, weight_kld = self._kld(mu, logvar, epoch)
        loss = loss_rec + weight_kld * loss_kld
        return loss, torch.stack([loss.detach(), loss_rec.detach(), loss_kld.detach()]).cpu().numpy()

    def calc_loss_rollout(self, data, epoch, eps_list: Optional[List[torch.Tensor]] = None):
        """Loss over consecutive 20-frame primitives of one long sequence (:435-505).  Primitive k covers frames
        [k t_pred, k t_pred + 20) and lives in the frame of its own first body; from the second primitive on, the motion seed
        is the model's reconstruction of the previous one, carried over between the two frames.  All canonical frames come from
        one kernel launch on the stacked first-frame joints; the loss is the mean over primitives."""
        ref_markers, ref_jts = data
        n_t, n_b = ref_markers.shape[:2]
        t_his = self.t_his
        t_pred = 20 - t_his
        starts = [s for s in range(0, n_t, t_pred) if s + 20 < n_t][:self.max_rollout]
        joints0 = ref_jts.reshape(n_t, n_b, -1, 3)[starts]                                   # [K,b,J,3]
        R_all, T_all = canonical_frame(joints0.reshape(len(starts) * n_b, -1, 3))
        R_all, T_all = R_all.view(len(starts), n_b, 3, 3), T_all.view(len(starts), n_b, 1, 3)
        # rigid frame changes of [t,b,p,3] point sets, written as broadcast multiply + sum (3-term dot products, no batched GEMM)
        rot = lambda M, x: (M[None, :, None] * x.unsqueeze(-2)).sum(-1)                        # y_i = sum_j M[b,i,j] x_j
        into = lambda x, R, T: rot(R.transpose(1, 2), x - T.unsqueeze(0))                      # world -> frame: R^T (x - T)
        out_of = lambda x, R, T: rot(R, x) + T.unsqueeze(0)                                    # frame -> world
        losses, infos, Y_rec = [], [], None
        for k, s in enumerate(starts):
            window = ref_markers[s:s + 20, :, :self.model.in_dim]
            if k == 0:                                                    # the data is already in the frame of frame 0
                X, Y = window[:t_his], window[t_his:]
            else:
                Y = into(window[t_his:].reshape(t_pred, n_b, -1, 3), R_all[k], T_all[k]).reshape(t_pred, n_b, -1)
                seed_world = out_of(Y_rec[-t_his:].reshape(t_his, n_b, -1, 3), R_all[k - 1], T_all[k - 1])
                X = into(seed_world, R_all[k], T_all[k]).reshape(t_his, n_b, -1)
            X, Y = X.detach().contiguous(), Y.detach().contiguous()
            Y_rec, mu, logvar = self.model.forward_train(X, Y, None if eps_list is None else eps_list[k])
            rec = self._calc_loss_rec(Y, Y_rec)
            kld, w_kld = self._kld(mu, logvar, epoch)
            losses.append(rec + w_kld * kld)
            infos.append(torch.stack([losses[-1].detach(), rec.detach(), kld.detach()]))
        return torch.stack(losses).mean(), torch.stack(infos).mean(0).cpu().numpy()

    def step(self, optimizer, loss):
        """loss.backward() + optimizer.step() around the flat gradient buffer (:541-553)."""
        self.grads.zero()
        loss.backward(retain_graph=False)
        optimizer.step()

    # ---- training loop (:509-591) -------------------------------------------------------------------------------------
    def train(self, batch_gen):
        self.build_model()
        tc = self.trainconfig
        starting_epoch = 0
        optimizer = torch.optim.Adam(self.model.parameters(), lr=tc["learning_rate"])
        scheduler = get_scheduler(optimizer, policy="lambda", num_epochs_fix=tc["num_epochs_fix"], num_epochs=tc["num_epochs"])
        if tc.get("resume_training", False):
            ckp_list = sorted(glob.glob(os.path.join(tc["save_dir"], "epoch-*.ckp")), key=os.path.getmtime)
            if not ckp_list:
                raise FileExistsError("the pre-trained checkpoint does not exist.")
            checkpoint = torch.load(ckp_list[-1], map_location=self.device)
            self.model.load_state_dict(checkpoint["model_state_dict"])
            if not tc.get("fine_tune", False):
                optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
                starting_epoch = checkpoint["epoch"]
        loss_names = ["ALL", "REC", "KLD"]
        history = []
        for epoch in range(starting_epoch, tc["num_epochs"]):
            epoch_losses, epoch_nsamples = 0, 0
            stime = time.time()
            while batch_gen.has_next_rec():
                if self.max_rollout is None:
                    data = batch_gen.next_batch(tc["batch_size"], noise=tc.get("noise"))
                    if data is None:
                        continue
                    loss, items = self.calc_loss(data.to(self.device), epoch)
                else:
                    data = batch_gen.next_batch_with_jts(tc["batch_size"], noise=tc.get("noise"))
                    if data is None:
                        continue
                    loss, items = self.calc_loss_rollout(data, epoch)
                self.step(optimizer, loss)
                epoch_losses = epoch_losses + items
                epoch_nsamples += 1
            batch_gen.reset() if self.max_rollout is None else batch_gen.reset_with_jts()
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
# canonicalised motion-primitive files and the batch generator over them
# ---------------------------------------------------------------------------------------------------------------------
PRIMITIVE_KEYS = ("trans", "poses", "betas", "gender", "mocap_framerate", "transf_rotmat", "transf_transl", "marker_cmu_41",
                  "marker_ssm2_67", "joints", "pelvis_loc")


def write_canonicalized_primitive(path, trans, poses, betas, gender, marker_ssm2_67, joints, transf_rotmat=None, transf_transl=None,
                                  marker_cmu_41=None, mocap_framerate=120.0):
    """One `subseq_XXXXX.npz` of the canonicalised AMASS / SAMP sets (written upstream by utils_canonicalize_samp.py, read by
    batch_gen_amass.py:152-170 and environments.py:188-194,480-483): T frames in the frame of frame 0."""
    T = len(trans)
    joints = np.asarray(joints, np.float64).reshape(T, -1)
    np.savez(path, trans=np.asarray(trans, np.float64), poses=np.asarray(poses, np.float64), betas=np.asarray(betas, np.float64),
             gender=np.asarray(gender), mocap_framerate=np.float64(mocap_framerate),
             transf_rotmat=np.eye(3) if transf_rotmat is None else np.asarray(transf_rotmat, np.float64),
             transf_transl=np.zeros((1, 3)) if transf_transl is None else np.asarray(transf_transl, np.float64),
             marker_cmu_41=np.zeros((T, 41, 3)) if marker_cmu_41 is None else np.asarray(marker_cmu_41, np.float64),
             marker_ssm2_67=np.asarray(marker_ssm2_67, np.float64).reshape(T, 67, 3), joints=joints,
             pelvis_loc=joints.reshape(T, -1, 3)[:, 0])


class BatchGeneratorAMASSCanonicalized:
    """batch_gen_amass.py:61-270 for body_repr 'ssm2_67' with everything read to RAM (the configuration
    train_GAMMAPredictor.py uses): `<amass_data_path>/<subset>/*.npz` -> data_all[b,t,201], jts_all[b,t,22,3] on the device."""

    def __init__(self, amass_data_path, amass_subset_name=None, sample_rate=3, body_repr="ssm2_67", read_to_ram=True, device="cuda"):
        if body_repr != "ssm2_67" or not read_to_ram:
            raise NotImplementedError("only body_repr='ssm2_67' with read_to_ram=True is used by the marker predictor")
        self.rec_list, self.index_rec = [], 0
        self.amass_data_path, self.amass_subset_name, self.sample_rate = amass_data_path, amass_subset_name, sample_rate
        self.body_repr, self.device = body_repr, device
        self.max_len = 200 if "x10" in amass_data_path else 20
        self.data_all = self.jts_all = None

    def get_rec_list(self, shuffle_seed=None, to_gpu=True):
        if self.amass_subset_name is not None:
            self.rec_list = []
            for subset in self.amass_subset_name:
                self.rec_list += sorted(glob.glob(os.path.join(self.amass_data_path, subset, "*.npz")))
        else:
            self.rec_list = sorted(glob.glob(os.path.join(self.amass_data_path, "*/*.npz")))
        (random.Random(shuffle_seed) if shuffle_seed is not None else random).shuffle(self.rec_list)
        data, jts = [], []
        for rec in self.rec_list:
            with np.load(rec) as d:
                if float(d["mocap_framerate"]) != 120:
                    continue
                sr = self.sample_rate
                pose, transl = d["poses"][::sr, :66], d["trans"][::sr]
                if not (np.isfinite(pose).all() and np.isfinite(transl).all()):
                    continue
                mk = d["marker_ssm2_67"][::sr][:self.max_len]
                jt = d["joints"][::sr].reshape([-1, 22, 3])[:self.max_len]
            if len(mk) < self.max_len:
                continue
            data.append(mk.reshape(-1, 67 * 3))
            jts.append(jt)
        if not data:
            raise FileNotFoundError(f"no usable canonicalised primitives under {self.amass_data_path}")
        dev = self.device if to_gpu else "cpu"
        self.data_all = torch.tensor(np.stack(data), dtype=torch.float32, device=dev)   # [b,t,d]
        self.jts_all = torch.tensor(np.stack(jts), dtype=torch.float32, device=dev)     # [b,t,22,3]

    def has_next_rec(self):
        return self.index_rec < self.data_all.shape[0]

    def _permute(self):
        idx = torch.randperm(self.data_all.shape[0], device=self.data_all.device)
        self.data_all, self.jts_all = self.data_all[idx], self.jts_all[idx]
        self.index_rec = 0

    reset = _permute
    reset_with_jts = _permute

    def next_batch(self, batch_size=64, noise=None):
        if noise is not None:
            # batch_gen_amass.py:226-259 re-poses the body with a noisy orientation, but reads the poses from `pose_all`, which
            # get_rec_list fills with the MARKER features (:208), so upstream that branch produces no usable bodies; no shipped
            # config sets `noise` (train_GAMMAPredictor.py:49 has it commented out).  Nothing to reproduce.
            raise NotImplementedError("next_batch(noise=...): the reference's augmentation branch is not usable upstream (see comment)")
        out = self.data_all[self.index_rec:self.index_rec + batch_size]
        self.index_rec += batch_size
        return out.permute(1, 0, 2).contiguous().to(self.device)   # [t,b,d]

    def next_batch_with_jts(self, batch_size=64, noise=None):
        d = self.data_all[self.index_rec:self.index_rec + batch_size].permute(1, 0, 2).contiguous()
        j = self.jts_all[self.index_rec:self.index_rec + batch_size].permute(1, 0, 2, 3).contiguous()
        self.index_rec += batch_size
        return d.to(self.device), j.to(self.device)

    def get_all_data(self):
        return self.data_all.permute(1, 0, 2)
