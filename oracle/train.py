"""Marker-predictor (C-VAE) training loss, CPU restatement.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (reference file:line, relative to motion/):
  cvae_encode / cvae_forward      models/models_GAMMA_primitive.py:75-110  (GAMMAPrimitiveVAE.encode / forward; VAE._sample
                                  baseops.py:650-653 with the noise `eps` passed in instead of drawn)
  loss_rec / predictor_loss       models/models_GAMMA_primitive.py:401-432 (GAMMAPrimitiveVAETrainOP._calc_loss_rec / calc_loss)
  predictor_loss_rollout          models/models_GAMMA_primitive.py:435-505 (calc_loss_rollout: canonical frames re-derived per
                                  primitive from the reference joints, the motion seed of primitive n > 0 is the model's own
                                  reconstruction carried into the new frame)
Pinned by tests/golden/predictor_train_ref.npz (scripts/gen_goldens.py runs the reference's TrainOP on the CPU).
Everything is plain differentiable torch over a reference-keyed state dict, so `torch.autograd.grad` gives the oracle
gradients the HIP-backed training step is compared with.
"""
import torch

from . import nets
from .env import get_new_coordinate

LOSS_CFG = {"weight_rec": 1.0, "weight_td": 3.0, "weight_kld": 1.0, "annealing_kld": False, "robust_kld": True}


def cvae_encode(sd, x, y, prefix=""):
    hx = nets.gru_last_hidden(x, sd, prefix + "x_enc")
    hy = nets.gru_last_hidden(y, sd, prefix + "e_rnn")
    h = nets.mlp(torch.cat([hx, hy], dim=-1), sd, prefix + "e_mlp", 2, torch.tanh)
    return nets.linear(h, sd, prefix + "e_mu"), nets.linear(h, sd, prefix + "e_logvar")


def cvae_forward(sd, x, y, eps, prefix=""):
    """x[t_his,b,201], y[t_pred,b,201], eps[b,128] -> y_pred, mu, logvar."""
    mu, logvar = cvae_encode(sd, x, y, prefix)
    z = mu + eps * torch.exp(0.5 * logvar)
    return nets.cvae_decode(sd, x, z, t_pred=y.shape[0], prefix=prefix), mu, logvar


def loss_rec(Y, Y_rec, cfg=LOSS_CFG):
    l1 = (Y - Y_rec).abs().mean()
    td = ((Y_rec[1:] - Y_rec[:-1]) - (Y[1:] - Y[:-1])).abs().mean()
    return cfg["weight_rec"] * l1 + cfg["weight_td"] * td


def _kld(mu, logvar, cfg, epoch, num_epochs):
    kld = 0.5 * torch.mean(-1 - logvar + mu.pow(2) + logvar.exp())
    if cfg["robust_kld"]:
        kld = torch.sqrt(1 + kld ** 2) - 1
    w = cfg["weight_kld"]
    if cfg["annealing_kld"]:
        w = min(float(epoch) / (0.9 * num_epochs), 1.0) * cfg["weight_kld"]
    return kld, w


def predictor_loss(sd, data, eps, t_his=2, cfg=LOSS_CFG, epoch=0, num_epochs=400, prefix=""):
    """calc_loss: data[t,b,201] -> loss, (loss, rec, kld)."""
    X, Y = data[:t_his], data[t_his:, :, :201]
    Y_rec, mu, logvar = cvae_forward(sd, X, Y, eps, prefix)
    rec = loss_rec(Y, Y_rec, cfg)
    kld, w = _kld(mu, logvar, cfg, epoch, num_epochs)
    loss = rec + w * kld
    return loss, (loss, rec, kld)


def predictor_loss_rollout(sd, ref_markers, ref_jts, eps_list, t_his=2, max_rollout=8, cfg=LOSS_CFG, epoch=0, num_epochs=400,
                           prefix=""):
    """calc_loss_rollout: ref_markers[n_t,n_b,201], ref_jts[n_t,n_b,J*3]; eps_list[i][n_b,128] per primitive."""
    n_t, n_b = ref_markers.shape[:2]
    jts = ref_jts.reshape(n_t, n_b, -1, 3)
    t, losses, infos = 0, [], []
    Y_rec = R_prev = T_prev = None
    while t < n_t:
        t_ub = t + 20
        if t_ub >= n_t:
            break
        t_pred = 20 - t_his
        mk, jt = ref_markers[t:t_ub], jts[t:t_ub]
        if t == 0:
            X = mk[:t_his].detach()
            Y = mk[t_his:, :, :201].detach()
            R_prev, T_prev = get_new_coordinate(jt[0])
        else:
            R_cur, T_cur = get_new_coordinate(jt[0])
            Yg = mk[t_his:, :, :201].reshape(t_pred, n_b, -1, 3)
            Y = torch.einsum("bij,tbpj->tbpi", R_cur.permute(0, 2, 1), Yg - T_cur.unsqueeze(0))
            X_prev = Y_rec[-t_his:].reshape(t_his, n_b, -1, 3)
            Xg = torch.einsum("bij,tbpj->tbpi", R_prev, X_prev) + T_prev.unsqueeze(0)
            X = torch.einsum("bij,tbpj->tbpi", R_cur.permute(0, 2, 1), Xg - T_cur.unsqueeze(0))
            Y = Y.reshape(t_pred, n_b, -1).detach()
            X = X.reshape(t_his, n_b, -1).detach()
            R_prev, T_prev = R_cur, T_cur
        Y_rec, mu, logvar = cvae_forward(sd, X, Y, eps_list[len(losses)], prefix)
        rec = loss_rec(Y, Y_rec, cfg)
        kld, w = _kld(mu, logvar, cfg, epoch, num_epochs)
        losses.append(rec + w * kld)
        infos.append(torch.stack([losses[-1].detach(), rec.detach(), kld.detach()]))
        t += t_pred
        if len(losses) >= max_rollout:
            break
    loss = torch.stack(losses).mean()
    return loss, torch.stack(infos).mean(0)


# ---- body-regressor training loss (models/models_GAMMA_primitive.py:617-633, GAMMARegressorTrainOP.calc_loss) -------------
# Pinned by tests/golden/regressor_train_ref.npz (scripts/gen_goldens.py regressor_train runs the reference class with its
# body model replaced by the adapter around smplx_lbs.py and torchgeometry by rot.py: the class's own arithmetic is pinned,
# the two third-party packages stay restated).

def regressor_marker_loss(bm, marker_ids, x_ref, xb, betas, weight_reg_hpose=0.01):
    """x_ref[n,67,3], xb[n,93] (axis-angle), betas[n,10] -> loss, (marker L1, hand-pose mean square)."""
    from .smplx_lbs import smplx_forward
    x_pred = smplx_forward(bm, xb, betas)[0][:, marker_ids]
    loss_marker = (x_ref.to(x_pred.dtype) - x_pred).abs().mean()
    loss_hpose = (xb[:, 69:] ** 2).mean()
    return loss_marker + weight_reg_hpose * loss_hpose, (loss_marker, loss_hpose)


def regressor_loss(sd, bm, marker_ids, marker_ref, betas, weight_reg_hpose=0.01, prefix=""):
    """One loop body of GAMMARegressorTrainOP.train (:670-678): MoshRegressor.forward on the reference markers, then calc_loss."""
    xb = nets.regressor_forward(sd, marker_ref.reshape(marker_ref.shape[0], -1), betas, prefix=prefix)
    loss, items = regressor_marker_loss(bm, marker_ids, marker_ref, xb, betas, weight_reg_hpose)
    return loss, items, xb
