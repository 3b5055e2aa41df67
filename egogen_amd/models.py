"""Network containers of the hot path with the reference's module / state_dict layout, executed by the
HIP library for inference.

Mirrors (names, attribute structure and therefore state_dict keys are the reference's, so its
checkpoints load unchanged):
  models/baseops.py:615-641                     MLP
  models/models_GAMMA_primitive.py:36-101        GAMMAPrimitiveVAE   (decode / sample_prior)
  models/models_GAMMA_primitive.py:160-301       ResNetBlock, MoshRegressor
  models/models_GAMMA_primitive.py:307-360       GAMMAPrimitiveCombo (sample_prior)
  models/models_policy_ppo.py:24-39,233-358      MLPBlock, GAMMAPolicyBase, GAMMAActor, GAMMACritic, ActorCritic
  human_body_prior VPoser v1 encoder [upstream]  VPoserEncoder

The torch modules only own the parameters (and provide the autograd forward used by the PPO update);
every rollout-time forward goes through libegogen_hip.so and raises if tensors are not on the GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

FUSED_UPDATE_OPS = True  # HIP-backed autograd nodes for GRU gates / posenc on the GPU (tests flip it to compare)

_ACT = {"tanh": torch.tanh, "relu": torch.relu, "lrelu": lambda x: F.leaky_relu(x, 0.01)}


def _p(t: torch.Tensor) -> int:
    assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32, "HIP path needs contiguous fp32 cuda tensors"
    return t.data_ptr()


def pack3(W: torch.Tensor, col0: int = 0, ncols: Optional[int] = None) -> torch.Tensor:
    """Packed image (three bf16 planes in MFMA fragment order, `egx_pack3`) of the columns [col0, col0 + ncols) of a
    row-major fp32 device matrix; returned as an opaque byte tensor."""
    lib = _lib.load()
    W = W.to(torch.float32)
    if W.stride(-1) != 1:
        W = W.contiguous()
    R, K = int(W.shape[0]), int(W.shape[1] - col0 if ncols is None else ncols)
    buf = torch.zeros(lib.egx_pack3_bytes(R, K), dtype=torch.uint8, device=W.device)
    _lib.check(lib.egx_pack3(C.c_void_p(W.data_ptr()), R, K, int(W.stride(0)), int(col0), _lib.ptr(buf), (K + 31) // 32, 0,
                             _lib.current_stream_ptr()), "egx_pack3")
    return buf


class _Workspace:
    """Scratch of the network entry points, one buffer per stream (calls on distinct streams never share scratch)."""

    def __init__(self):
        self.bufs: Dict[int, torch.Tensor] = {}
        self._retired = []  # outgrown buffers stay alive: a captured HIP graph may still hold their addresses

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != torch.device(device):
            if buf is not None:
                self._retired.append(buf)
            buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


class MLP(nn.Module):
    def __init__(self, in_dim, h_dims=(128, 128), activation="tanh"):
        super().__init__()
        self.act_name = activation
        self.out_dim = h_dims[-1]
        self.layers = nn.ModuleList()
        d = in_dim
        for h in h_dims:
            self.layers.append(nn.Linear(d, h))
            d = h

    def forward(self, x):
        for fc in self.layers:
            x = _ACT[self.act_name](fc(x))
        return x


class GAMMAPrimitiveVAE(nn.Module):
    """Marker predictor.  Only the prior-sampling half (decode) is on the crowd_ppo path; the encoder
    parameters exist so that reference checkpoints (`epoch-400.ckp`) load with strict=True."""

    def __init__(self, configs):
        super().__init__()
        if configs["body_repr"] != "ssm2_67":
            raise NotImplementedError("only body_repr ssm2_67 is used by crowd_ppo")
        self.in_dim = in_dim = 67 * 3
        self.h_dim = h = configs["h_dim"]
        self.z_dim = z = configs["z_dim"]
        hd = list(configs["hdims_mlp"])
        assert (h, z, hd, configs["use_drnn_mlp"], configs["residual"]) == (256, 128, [512, 256], True, True), \
            "HIP decode is specialised to the released MPVAE_samp20_2frame_rollout config"
        self.x_enc = nn.GRU(in_dim, h)
        self.e_rnn = nn.GRU(in_dim, h)
        self.e_mlp = MLP(2 * h, hd, "tanh")
        self.e_mu = nn.Linear(self.e_mlp.out_dim, z)
        self.e_logvar = nn.Linear(self.e_mlp.out_dim, z)
        self.drnn_mlp = MLP(h, hd + [h], "tanh")
        self.d_rnn = nn.GRUCell(in_dim + z + h, h)
        self.d_mlp = MLP(h, hd, "tanh")
        self.d_out = nn.Linear(self.d_mlp.out_dim, in_dim)

    def forward_train(self, x: torch.Tensor, y: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """GAMMAPrimitiveVAE.forward (models_GAMMA_primitive.py:103-110: encode, VAE._sample, decode) for TRAINING, as an
        autograd graph of HIP-backed nodes (fused_ops.LinearFn / GRUSeqFn / GRUPointwiseFn: weight gradients accumulate into
        the flat buffer of `fused_ops.FlatGrads`, so the 18 uses of the decoder cell's weights cost no AccumulateGrad
        kernels).  x[t_his,b,201], y[t_pred,b,201], eps[b,z] (drawn here when None) -> y_pred[t_pred,b,201], mu, logvar."""
        from .fused_ops import GRUPointwiseFn, linear_act, linear_fn
        if not x.is_cuda:
            raise _lib.EgxError("the training forward runs on the HIP device only (no CPU fallback)")
        t_his, nb, _ = x.shape
        t_pred = y.shape[0]
        x = x.to(torch.float32).contiguous()
        y = y.to(torch.float32).contiguous()
        # encode (:75-80).  decode (:85) evaluates x_enc on the same x again: the same tensor, used twice.
        hx = _gru_last_fused_tm(self.x_enc, x.reshape(t_his * nb, -1), t_his)
        hy = _gru_last_fused_tm(self.e_rnn, y.reshape(t_pred * nb, -1), t_pred)
        h = torch.cat([hx, hy], dim=-1)
        for fc in self.e_mlp.layers:
            h = linear_act(h, fc, "tanh")
        mu, logvar = linear_act(h, self.e_mu), linear_act(h, self.e_logvar)
        if eps is None:
            eps = torch.randn_like(mu)
        z = mu + eps * torch.exp(0.5 * logvar)                     # VAE._sample, baseops.py:650-653
        # decode (:83-101)
        h_rnn = hx
        for fc in self.drnn_mlp.layers:
            h_rnn = linear_act(h_rnn, fc, "tanh")
        ys = []
        y_i = x[-1][:, :self.in_dim]
        cell = self.d_rnn
        for _ in range(t_pred):
            y_p = y_i
            rnn_in = torch.cat([hx, z, y_p], dim=-1)
            gi = linear_fn(rnn_in, cell.weight_ih, cell.bias_ih)
            gh = linear_fn(h_rnn, cell.weight_hh, cell.bias_hh)
            h_rnn = GRUPointwiseFn.apply(gi, gh, h_rnn)
            hfc = h_rnn
            for fc in self.d_mlp.layers:
                hfc = linear_act(hfc, fc, "tanh")
            y_i = linear_act(hfc, self.d_out) + y_p                # residual
            ys.append(y_i)
        return torch.stack(ys), mu, logvar


class ResNetBlock(nn.Module):
    def __init__(self, in_dim, h_dim, out_dim, n_blocks, actfun="relu"):
        super().__init__()
        self.in_fc = nn.Linear(in_dim, h_dim)
        self.layers = nn.ModuleList([MLP(h_dim, (h_dim, h_dim), actfun) for _ in range(n_blocks)])
        self.out_fc = nn.Linear(h_dim, out_dim)


class MoshRegressor(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert (config["h_dim"], config["n_blocks"], config["n_recur"], config["actfun"], config.get("use_cont", False)) == \
            (128, 10, 3, "relu", True), "HIP regressor is specialised to the released MoshRegressor_v3 config"
        self.in_dim = 67 * 3
        self.body_dim = 3 + 6 + 21 * 6 + 24
        self.pnet = ResNetBlock(self.in_dim + self.body_dim + 10, 128, self.body_dim, 10, "relu")



class GAMMAPrimitiveCombo(nn.Module):
    def __init__(self, markercfg, bparamscfg):
        super().__init__()
        self.predictor = GAMMAPrimitiveVAE(markercfg)
        self.regressor = MoshRegressor(bparamscfg)
        self._ws = _Workspace()
        self._wstruct = None
        self._wkey = None
        self._gen = 0   # bumped whenever the parameters change (mark_dirty / load_state_dict / _apply): the packed images are copies

    def mark_dirty(self):
        """The parameters changed: re-derive the folded / packed weight images before the next `sample_prior`.  Called by
        load_state_dict and by .to() / .cuda() / .float(); in-place edits torch versions (`optimizer.step()`, `p.copy_()`) are
        noticed through the tensors' version counters; a kernel that writes the parameters BY ADDRESS must call this itself."""
        self._gen += 1

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self.mark_dirty()
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.mark_dirty()
        return out

    def _weights(self) -> _lib.PriorWeights:
        p, r = self.predictor, self.regressor.pnet
        # in-place edits torch can see (optimizer.step(), copy_, a sub-module's load_state_dict) bump a tensor's _version; the sum
        # over a cached parameter list costs a few microseconds per call (the generator `self.parameters()` walked the module
        # tree on every env step).  Writes by ADDRESS are invisible here: they need mark_dirty().
        plist = self.__dict__.get("_plist")
        if plist is None or self.__dict__.get("_plist_gen") != self._gen:
            plist = list(self.parameters())
            self.__dict__["_plist"], self.__dict__["_plist_gen"] = plist, self._gen
        key = (p.x_enc.weight_ih_l0.data_ptr(), r.out_fc.weight.data_ptr(), self._gen, sum(t._version for t in plist))
        if self._wstruct is not None and self._wkey == key:
            return self._wstruct
        w = _lib.PriorWeights()
        w.x_enc_w_ih, w.x_enc_w_hh = _p(p.x_enc.weight_ih_l0), _p(p.x_enc.weight_hh_l0)
        w.x_enc_b_ih, w.x_enc_b_hh = _p(p.x_enc.bias_ih_l0), _p(p.x_enc.bias_hh_l0)
        for i in range(3):
            w.drnn_w[i], w.drnn_b[i] = _p(p.drnn_mlp.layers[i].weight), _p(p.drnn_mlp.layers[i].bias)
        w.d_rnn_w_ih, w.d_rnn_w_hh = _p(p.d_rnn.weight_ih), _p(p.d_rnn.weight_hh)
        w.d_rnn_b_ih, w.d_rnn_b_hh = _p(p.d_rnn.bias_ih), _p(p.d_rnn.bias_hh)
        for i in range(2):
            w.d_mlp_w[i], w.d_mlp_b[i] = _p(p.d_mlp.layers[i].weight), _p(p.d_mlp.layers[i].bias)
        w.d_out_w, w.d_out_b = _p(p.d_out.weight), _p(p.d_out.bias)
        w.reg_in_w, w.reg_in_b = _p(r.in_fc.weight), _p(r.in_fc.bias)
        for b in range(10):
            for k in range(2):
                w.reg_blk_w[2 * b + k], w.reg_blk_b[2 * b + k] = _p(r.layers[b].layers[k].weight), _p(r.layers[b].layers[k].bias)
        w.reg_out_w, w.reg_out_b = _p(r.out_fc.weight), _p(r.out_fc.bias)
        # output layer folded into the GRU cell's input product (egx_prior_weights.d_comb_*), in float64
        with torch.no_grad():
            wy = p.d_rnn.weight_ih[:, p.d_rnn.weight_ih.shape[1] - p.d_out.weight.shape[0]:].double()
            self._comb_w = (wy @ p.d_out.weight.double()).float().contiguous()
            self._comb_b = (wy @ p.d_out.bias.double()).float().contiguous()
        w.d_comb_w, w.d_comb_b = _p(self._comb_w), _p(self._comb_b)
        # the decoder's and the regressor's dense weights as three bf16 planes in MFMA fragment order (egx_prior_packed3,
        # csrc/dense3.hip): what egx_sample_prior computes from
        if p.x_enc.weight_ih_l0.is_cuda:
            self._p3_bufs = {}
            p3 = _lib.PriorPacked3()
            wih = p.d_rnn.weight_ih
            jobs = {"x_enc_w_ih": (p.x_enc.weight_ih_l0, 0, None), "x_enc_w_hh": (p.x_enc.weight_hh_l0, 0, None),
                    "d_rnn_w_hz": (wih, 0, 384), "d_rnn_w_y": (wih, 384, wih.shape[1] - 384), "d_rnn_w_hh": (p.d_rnn.weight_hh, 0, None),
                    "d_comb_w": (self._comb_w, 0, None), "d_out_w": (p.d_out.weight, 0, None)}
            for i in range(3):
                jobs[f"drnn_w{i}"] = (p.drnn_mlp.layers[i].weight, 0, None)
            for i in range(2):
                jobs[f"d_mlp_w{i}"] = (p.d_mlp.layers[i].weight, 0, None)
            for name, (W, col0, ncols) in jobs.items():
                self._p3_bufs[name] = pack3(W.detach(), col0, ncols)
            for name in ("x_enc_w_ih", "x_enc_w_hh", "d_rnn_w_hz", "d_rnn_w_y", "d_rnn_w_hh", "d_comb_w", "d_out_w"):
                setattr(p3, name, self._p3_bufs[name].data_ptr())
            for i in range(3):
                p3.drnn_w[i] = self._p3_bufs[f"drnn_w{i}"].data_ptr()
            for i in range(2):
                p3.d_mlp_w[i] = self._p3_bufs[f"d_mlp_w{i}"].data_ptr()
            lib = _lib.load()
            win = r.in_fc.weight.detach()
            self._p3_bufs["reg_in_m"] = pack3(win, 0, 201)
            self._p3_bufs["reg_in_xb"] = pack3(win, 201, 159)
            self._p3_bufs["reg_in_betas"] = pack3(win, 360, 10)
            self._p3_bufs["reg_out"] = pack3(r.out_fc.weight.detach())
            blk = [r.layers[b].layers[k] for b in range(10) for k in range(2)]
            per = lib.egx_pack3_bytes(128, 128)
            allb = torch.zeros(20 * per, dtype=torch.uint8, device=win.device)
            for i, fc in enumerate(blk):
                allb[i * per:(i + 1) * per].copy_(pack3(fc.weight.detach()))
            self._p3_bufs["reg_blk"] = allb
            self._p3_bufs["reg_blk_b"] = torch.stack([fc.bias.detach().float() for fc in blk]).contiguous()
            for name in ("reg_in_m", "reg_in_xb", "reg_in_betas", "reg_blk", "reg_out", "reg_blk_b"):
                setattr(p3, name, self._p3_bufs[name].data_ptr())
            self._p3 = p3
            w.packed3 = C.pointer(p3)
        self._wstruct, self._wkey = w, key
        return w

    @torch.no_grad()
    def sample_prior_into(self, x0: torch.Tensor, x1: torch.Tensor, x_ld: int, betas: torch.Tensor, z: torch.Tensor,
                          out_Y: torch.Tensor, out_Yb: torch.Tensor):
        """Raw form used by the vector env: x0/x1 are views of the two history frames with row stride x_ld."""
        lib = _lib.load()
        A = int(z.shape[0])
        ws = self._ws.get(lib.egx_sample_prior_workspace_bytes(A), z.device)
        w = self._weights()
        rc = lib.egx_sample_prior(C.byref(w), C.c_void_p(x0.data_ptr()), C.c_void_p(x1.data_ptr()), int(x_ld),
                                  _lib.ptr(betas), _lib.ptr(z), A, _lib.ptr(out_Y), _lib.ptr(out_Yb),
                                  _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "egx_sample_prior")

    @torch.no_grad()
    def sample_prior(self, X: torch.Tensor, betas: torch.Tensor, z: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """models_GAMMA_primitive.py:334-360.  X[t_his=2,b,201], betas[18,b,10] (or [b,10]), z[b,128] or None
        -> Y_gen[18,b,201], Yb_gen[18,b,93] (axis-angle)."""
        if not X.is_cuda:
            raise _lib.EgxError("sample_prior runs on the HIP device only (no CPU fallback)")
        if X.dim() != 3 or X.shape[0] != 2 or X.shape[2] != 201:
            raise ValueError(f"X must be [2,b,201], got {tuple(X.shape)}")
        b = X.shape[1]
        if z is None:
            z = torch.randn(b, self.predictor.z_dim, device=X.device)
        X = X.to(torch.float32).contiguous()
        betas_a = (betas[0] if betas.dim() == 3 else betas).to(torch.float32).contiguous()
        z = z.to(torch.float32).contiguous()
        Y = torch.empty(18, b, 201, dtype=torch.float32, device=X.device)
        Yb = torch.empty(18, b, 93, dtype=torch.float32, device=X.device)
        self.sample_prior_into(X[0], X[1], 201, betas_a, z, Y, Yb)
        return Y, Yb


# ---------------------------------------------------------------------------------------------
# policy
# ---------------------------------------------------------------------------------------------

class MLPBlock(nn.Module):
    def __init__(self, h_dim, out_dim, n_blocks, actfun="relu", residual=True):
        super().__init__()
        self.residual = residual
        self.layers = nn.ModuleList([MLP(h_dim, (h_dim, h_dim), actfun) for _ in range(n_blocks)])
        self.out_fc = nn.Linear(h_dim, out_dim)

    def forward(self, x):
        h = x
        for layer in self.layers:
            h = layer(h) + (h if self.residual else 0)
        return self.out_fc(h)


_FREQ_CACHE: dict = {}


def _exact_freqs(L: int, device, dtype) -> torch.Tensor:
    """2^0..2^(L-1), built exactly on the host once per (device, dtype): a device-side pow() is not exact for 2^31 and
    sin/cos of x * 2^k are extremely sensitive to the last bit of the argument.  Cached so that no host-to-device copy
    happens inside a captured HIP graph."""
    key = (L, str(device), dtype)
    f = _FREQ_CACHE.get(key)
    if f is None:
        f = torch.ldexp(torch.ones(L, dtype=torch.float32), torch.arange(L)).to(device=device, dtype=dtype)
        _FREQ_CACHE[key] = f
    return f


def positional_encoding(x: torch.Tensor, L: int = 32) -> torch.Tensor:
    """models_policy_ppo.py:276-285: x[b,1] -> [b,2L], [sin(x 2^k), cos(x 2^k)] interleaved per k."""
    xf = x * _exact_freqs(L, x.device, x.dtype)  # [b,L]
    return torch.stack([torch.sin(xf), torch.cos(xf)], dim=-1).reshape(x.shape[0], 2 * L)


class GAMMAPolicyBase(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.h_dim = config["h_dim"]
        self.z_dim = config["z_dim"]
        if config["body_repr"] not in {"ssm2_67_condi_marker", "ssm2_67_condi_marker_map"}:
            raise NotImplementedError("other body_repr is not implemented yet.")
        self.in_dim = 67 * 3 * 2
        self.x_enc = nn.GRU(self.in_dim, self.h_dim)
        self.ego_enc = nn.GRU(32, self.h_dim)

    @staticmethod
    def _gru_last(gru: nn.GRU, x_seq: torch.Tensor) -> torch.Tensor:
        """nn.GRU over x_seq[t,b,in] from a zero state, written as explicit gate math (r,z,n) on F.linear so
        that the autograd path does not depend on the vendor RNN kernel; returns the last hidden state."""
        H = gru.hidden_size
        h = x_seq.new_zeros(x_seq.shape[1], H)
        fused = x_seq.is_cuda and FUSED_UPDATE_OPS
        for t in range(x_seq.shape[0]):
            gi = F.linear(x_seq[t], gru.weight_ih_l0, gru.bias_ih_l0)
            gh = F.linear(h, gru.weight_hh_l0, gru.bias_hh_l0)
            if fused:  # one HIP kernel forward, one backward, instead of ~15 + ~30 elementwise torch kernels
                from .fused_ops import GRUPointwiseFn
                h = GRUPointwiseFn.apply(gi, gh, h)
                continue
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
        return h

    def forward(self, obs):
        """autograd path (PPO update): obs dict -> hx[b,1152]."""
        nb = obs["state"].shape[0]
        hx = self._gru_last(self.x_enc, obs["state"].permute(1, 0, 2))
        he = self._gru_last(self.ego_enc, obs["egosensing"].permute(1, 0, 2))
        if obs["dist"].is_cuda and FUSED_UPDATE_OPS:
            from .fused_ops import posenc_dist_time
            pe = posenc_dist_time(obs["dist"].reshape(nb).float(), obs["time"].reshape(nb).float())
            return torch.cat([hx, he, pe], dim=-1)
        d = positional_encoding(obs["dist"].reshape(nb, 1))
        t = positional_encoding(obs["time"].reshape(nb, 1))
        return torch.cat([hx, he, d, t], dim=-1)


class GAMMAActor(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.h_dim, self.z_dim = config["h_dim"], config["z_dim"]
        self.min_logvar = config.get("min_logvar", -1)
        self.max_logvar = config.get("max_logvar", 3)
        self.pnet = MLPBlock(self.h_dim * 2 + 128, self.z_dim * 2, config["n_blocks"], actfun=config["actfun"])

    def forward(self, hx, state=None, info={}):
        zp = self.pnet(hx)
        return (zp[:, :self.z_dim], zp[:, self.z_dim:]), state


class GAMMACritic(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.h_dim, self.z_dim = config["h_dim"], config["z_dim"]
        self.vnet = MLPBlock(self.h_dim * 2 + 128, 1, config["n_blocks"], actfun=config["actfun"])

    def forward(self, hx, state=None, info={}):
        return self.vnet(hx)


class ActorCritic(nn.Module):
    def __init__(self, actor, critic, shared_net=None):
        super().__init__()
        self.actor = actor
        self.critic = critic
        if shared_net is not None:
            self.shared_net = shared_net


_TWO_STREAM_UPDATE = os.environ.get("EGX_TWO_STREAM_UPDATE", "1") == "1"
_SIDE_STREAMS: dict = {}


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[device] = s
    return s


def _gru_last_fused(gru: nn.GRU, x: torch.Tensor) -> torch.Tensor:
    """nn.GRU over x[b,T,in] from a zero state, last hidden state.  The input products of all T steps are ONE GEMM
    (each weight is used once per minibatch: its gradient is written by a single accumulate-GEMM), the gate math is the
    fused HIP node, and the first step's recurrent term is the bias alone (h_0 = 0)."""
    from .fused_ops import GRUSeqFn
    nb, T, _ = x.shape
    X = x.permute(1, 0, 2).reshape(T * nb, x.shape[2])
    w_ih, b_ih, w_hh, b_hh = gru.weight_ih_l0, gru.bias_ih_l0, gru.weight_hh_l0, gru.bias_hh_l0
    if any(t.grad is None for t in (w_ih, b_ih, w_hh, b_hh)):
        raise _lib.EgxError("GRUSeqFn needs pre-allocated gradient views (GAMMAPPOPolicy._ensure_flat_grads)")
    return GRUSeqFn.apply(X, w_ih, b_ih, w_hh, b_hh, w_ih.grad, b_ih.grad, w_hh.grad, b_hh.grad, T)


def _gru_last_fused_tm(gru: nn.GRU, x_tm: torch.Tensor, T: int) -> torch.Tensor:
    """As _gru_last_fused for an input that is already time-major and flattened: x_tm[T*b, in]."""
    from .fused_ops import GRUSeqFn
    w_ih, b_ih, w_hh, b_hh = gru.weight_ih_l0, gru.bias_ih_l0, gru.weight_hh_l0, gru.bias_hh_l0
    if any(t.grad is None for t in (w_ih, b_ih, w_hh, b_hh)):
        raise _lib.EgxError("GRUSeqFn needs pre-allocated gradient views (fused_ops.FlatGrads)")
    return GRUSeqFn.apply(x_tm, w_ih, b_ih, w_hh, b_hh, w_ih.grad, b_ih.grad, w_hh.grad, b_hh.grad, T)


def _mlpblock_fused(block: MLPBlock, x: torch.Tensor) -> torch.Tensor:
    from .fused_ops import ACT_CODE, linear_act, res_mlp
    h = x
    for mlp in block.layers:
        if block.residual and len(mlp.layers) == 2 and ACT_CODE[mlp.act_name] != 0 and mlp.layers[1].out_features == h.shape[1]:
            h = res_mlp(h, mlp.layers[0], mlp.layers[1], mlp.act_name)
            continue
        t = h
        last = len(mlp.layers) - 1
        for i, fc in enumerate(mlp.layers):
            t = linear_act(t, fc, mlp.act_name, res=h if (block.residual and i == last) else None)
        h = t
    return linear_act(h, block.out_fc, "none")


def fused_update_forward(shared_net: "GAMMAPolicyBase", actor: "GAMMAActor", critic: "GAMMACritic", obs, packed: bool = False):
    """Forward of (shared_net, actor, critic) for the PPO update with every dense layer a `fused_ops.LinearFn` node
    (models_policy_ppo.py:287-350).  Returns mu[b,128], raw logvar[b,128], value[b] - or, with `packed`, the actor head's raw
    output zp[b,256] = [mu | logvar] and value[b]."""
    from .fused_ops import posenc_dist_time
    nb = obs["state"].shape[0]
    if _TWO_STREAM_UPDATE:  # the two encoders are independent as well
        cur = torch.cuda.current_stream()
        side = _side_stream(obs["state"].device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            he = _gru_last_fused(shared_net.ego_enc, obs["egosensing"].float())
        hx = _gru_last_fused(shared_net.x_enc, obs["state"].float())
        cur.wait_stream(side)
    else:
        hx = _gru_last_fused(shared_net.x_enc, obs["state"].float())
        he = _gru_last_fused(shared_net.ego_enc, obs["egosensing"].float())
    pe = posenc_dist_time(obs["dist"].reshape(nb).float(), obs["time"].reshape(nb).float())
    h = torch.cat([hx, he, pe], dim=-1)
    if _TWO_STREAM_UPDATE:
        # actor and critic heads are independent given h: run the critic on a side stream (forked / joined around it;
        # inside a graph capture this becomes a parallel branch, and autograd replays each backward node on the stream of
        # its forward) - two half-filling GEMM streams share the chip instead of running back to back
        cur = torch.cuda.current_stream()
        side = _side_stream(h.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            value = _mlpblock_fused(critic.vnet, h).flatten()
        zp = _mlpblock_fused(actor.pnet, h)
        cur.wait_stream(side)
    else:
        zp = _mlpblock_fused(actor.pnet, h)
        value = _mlpblock_fused(critic.vnet, h).flatten()
    if packed:
        return zp, value
    return zp[:, :actor.z_dim], zp[:, actor.z_dim:], value


class PolicyHipRunner:
    """Rollout-time forward of (shared_net, actor, critic) through egx_policy_forward."""

    def __init__(self, shared_net: GAMMAPolicyBase, actor: GAMMAActor, critic: GAMMACritic):
        assert shared_net.h_dim == 512 and actor.z_dim == 128 and len(actor.pnet.layers) == 2
        self.shared_net, self.actor, self.critic = shared_net, actor, critic
        self._ws = _Workspace()
        self._wstruct = None
        self._wkey = None
        self._p3, self._dirty, self._p3_ver = None, True, -1

    def _weights(self) -> _lib.PolicyWeights:
        s, a, c = self.shared_net, self.actor.pnet, self.critic.vnet
        key = (s.x_enc.weight_ih_l0.data_ptr(), a.out_fc.weight.data_ptr(), c.out_fc.weight.data_ptr())
        if self._wstruct is not None and self._wkey == key:
            return self._wstruct
        w = _lib.PolicyWeights()
        w.x_enc_w_ih, w.x_enc_w_hh, w.x_enc_b_ih, w.x_enc_b_hh = (_p(s.x_enc.weight_ih_l0), _p(s.x_enc.weight_hh_l0),
                                                                    _p(s.x_enc.bias_ih_l0), _p(s.x_enc.bias_hh_l0))
        w.ego_enc_w_ih, w.ego_enc_w_hh, w.ego_enc_b_ih, w.ego_enc_b_hh = (_p(s.ego_enc.weight_ih_l0), _p(s.ego_enc.weight_hh_l0),
                                                                            _p(s.ego_enc.bias_ih_l0), _p(s.ego_enc.bias_hh_l0))
        for b in range(2):
            for k in range(2):
                w.actor_w[2 * b + k], w.actor_b[2 * b + k] = _p(a.layers[b].layers[k].weight), _p(a.layers[b].layers[k].bias)
                w.critic_w[2 * b + k], w.critic_b[2 * b + k] = _p(c.layers[b].layers[k].weight), _p(c.layers[b].layers[k].bias)
        w.actor_out_w, w.actor_out_b = _p(a.out_fc.weight), _p(a.out_fc.bias)
        w.critic_out_w, w.critic_out_b = _p(c.out_fc.weight), _p(c.out_fc.bias)
        self._p3 = None
        if s.x_enc.weight_ih_l0.is_cuda:
            self._p3_src = [("x_enc_w_ih", None, s.x_enc.weight_ih_l0), ("x_enc_w_hh", None, s.x_enc.weight_hh_l0),
                            ("ego_enc_w_ih", None, s.ego_enc.weight_ih_l0), ("ego_enc_w_hh", None, s.ego_enc.weight_hh_l0),
                            ("actor_out_w", None, a.out_fc.weight), ("critic_out_w", None, c.out_fc.weight)]
            for b in range(2):
                for k in range(2):
                    self._p3_src.append(("actor_w", 2 * b + k, a.layers[b].layers[k].weight))
                    self._p3_src.append(("critic_w", 2 * b + k, c.layers[b].layers[k].weight))
            lib = _lib.load()
            self._p3_bufs = [torch.zeros(lib.egx_pack3_bytes(int(W.shape[0]), int(W.shape[1])), dtype=torch.uint8, device=W.device)
                             for _, _, W in self._p3_src]
            p3 = _lib.PolicyPacked3()
            for (name, idx, _), buf in zip(self._p3_src, self._p3_bufs):
                if idx is None:
                    setattr(p3, name, buf.data_ptr())
                else:
                    getattr(p3, name)[idx] = buf.data_ptr()
            self._p3 = p3
            w.packed3 = C.pointer(p3)
            self._dirty = True
        self._wstruct, self._wkey = w, key
        return w

    def adopt_packed(self, packed3: "_lib.PolicyPacked3", refresh=None):
        """Use weight images somebody else keeps current (the update's `egx_policy_train` handle re-makes them at the head of
        every train step and at the end of `learn()`): the runner stops packing its own.  `refresh`: called before a forward
        when torch saw an in-place edit of a weight since the last forward (load_state_dict, copy_) - edits the owner cannot
        know about."""
        w = self._weights()
        self._adopted = packed3            # keep the struct alive
        self._adopted_refresh = refresh
        self._adopted_src = [W for _, _, W in self._p3_src] if self._p3 is not None else []
        self._adopted_ver = sum(int(W._version) for W in self._adopted_src)
        w.packed3 = C.pointer(packed3)
        self._p3 = None

    def mark_dirty(self):
        """The parameters changed through a path torch does not version (the flat AdamW kernel writes them by address):
        the packed images are re-made before the next forward."""
        self._dirty = True

    def _refresh_packed(self):
        """Re-pack the dense weights (three bf16 planes in MFMA fragment order) when any parameter changed since the images
        were made: explicit mark (learn()), or an in-place edit torch saw (optim.step(), load_state_dict, copy_)."""
        if self._p3 is None:
            if getattr(self, "_adopted_refresh", None) is not None:
                ver = sum(int(W._version) for W in self._adopted_src)
                if ver != self._adopted_ver:
                    self._adopted_refresh()
                    self._adopted_ver = ver
            return
        ver = sum(int(W._version) for _, _, W in self._p3_src)
        if not self._dirty and ver == self._p3_ver:
            return
        lib, st = _lib.load(), _lib.current_stream_ptr()
        for (_, _, W), buf in zip(self._p3_src, self._p3_bufs):
            R, K = int(W.shape[0]), int(W.shape[1])
            _lib.check(lib.egx_pack3(C.c_void_p(W.data_ptr()), R, K, K, 0, _lib.ptr(buf), (K + 31) // 32, 0, st), "egx_pack3")
        self._dirty, self._p3_ver = False, ver

    @torch.no_grad()
    def forward(self, obs: Dict[str, torch.Tensor], want_actor=True, want_critic=True, out: Optional[dict] = None):
        lib = _lib.load()
        st = obs["state"]
        if not st.is_cuda:
            raise _lib.EgxError("policy inference runs on the HIP device only (no CPU fallback)")
        n = int(st.shape[0])
        f = lambda t: t.to(torch.float32).contiguous()
        st, ego, dist, time = f(st), f(obs["egosensing"]), f(obs["dist"]).reshape(n), f(obs["time"]).reshape(n)
        out = out if out is not None else {}
        if want_actor and "mu" not in out:
            out["mu"] = torch.empty(n, 128, dtype=torch.float32, device=st.device)
            out["logvar"] = torch.empty(n, 128, dtype=torch.float32, device=st.device)
        if want_critic and "value" not in out:
            out["value"] = torch.empty(n, dtype=torch.float32, device=st.device)
        ws = self._ws.get(lib.egx_policy_workspace_bytes(n), st.device)
        w = self._weights()
        # adopted weight images carry only the planes their consumers read at the time they were made
        # (egx_policy_train_refresh): a switch of the rollout arithmetic re-makes them
        prec = int(lib.egx_policy_get_precision())
        if prec != getattr(self, "_last_prec", prec) and self._p3 is None and getattr(self, "_adopted_refresh", None) is not None:
            self._adopted_refresh()
        self._last_prec = prec
        self._refresh_packed()
        rc = lib.egx_policy_forward(C.byref(w), _lib.ptr(st), _lib.ptr(ego), _lib.ptr(dist), _lib.ptr(time), n,
                                    _lib.ptr(out["mu"]) if want_actor else None,
                                    _lib.ptr(out["logvar"]) if want_actor else None,
                                    _lib.ptr(out["value"]) if want_critic else None,
                                    _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "egx_policy_forward")
        return out


# ---------------------------------------------------------------------------------------------
# VPoser v1 encoder
# ---------------------------------------------------------------------------------------------

class VPoserEncoder(nn.Module):
    """Encoder half of human_body_prior v1 VPoser (num_neurons=512, latentD=32) with its parameter names,
    so a `snapshot` state_dict loads with strict=False.  encode_mean() = vposer.encode(x).loc in eval mode."""

    def __init__(self, num_neurons=512, latentD=32, n_features=63):
        super().__init__()
        self.bodyprior_enc_bn1 = nn.BatchNorm1d(n_features)
        self.bodyprior_enc_fc1 = nn.Linear(n_features, num_neurons)
        self.bodyprior_enc_bn2 = nn.BatchNorm1d(num_neurons)
        self.bodyprior_enc_fc2 = nn.Linear(num_neurons, num_neurons)
        self.bodyprior_enc_mu = nn.Linear(num_neurons, latentD)
        self.bodyprior_enc_logvar = nn.Linear(num_neurons, latentD)
        self._folded = None

    @torch.no_grad()
    def fold(self):
        """Fold the eval-mode BatchNorms into the following Linear (float64 on the host)."""
        def affine(bn):
            s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            return s, bn.bias.double() - bn.running_mean.double() * s
        s1, t1 = affine(self.bodyprior_enc_bn1)
        s2, t2 = affine(self.bodyprior_enc_bn2)
        W1, b1 = self.bodyprior_enc_fc1.weight.double(), self.bodyprior_enc_fc1.bias.double()
        W2, b2 = self.bodyprior_enc_fc2.weight.double(), self.bodyprior_enc_fc2.bias.double()
        dev = self.bodyprior_enc_fc1.weight.device
        f = lambda t: t.float().contiguous().to(dev)
        self._folded = {"fc1_w": f(W1 * s1[None, :]), "fc1_b": f(b1 + W1 @ t1),
                        "fc2_w": f(W2 * s2[None, :]), "fc2_b": f(b2 + W2 @ t2),
                        "mu_w": f(self.bodyprior_enc_mu.weight), "mu_b": f(self.bodyprior_enc_mu.bias)}
        w = _lib.VposerWeights()
        for k, v in self._folded.items():
            setattr(w, k, _p(v))
        if dev.type == "cuda":   # packed images of the folded weights: what egx_vposer_encode computes from
            self._packed = {k: pack3(self._folded[k]) for k in ("fc1_w", "fc2_w", "mu_w")}
            for k, v in self._packed.items():
                setattr(w, k + "3", v.data_ptr())
        self._wstruct = w
        return self

    @torch.no_grad()
    def encode_mean_into(self, x: torch.Tensor, x_ld: int, n: int, out: torch.Tensor):
        lib = _lib.load()
        if self._folded is None:
            self.fold()
        rc = lib.egx_vposer_encode(C.byref(self._wstruct), C.c_void_p(x.data_ptr()), int(x_ld), int(n), _lib.ptr(out),
                                   None, 0, _lib.current_stream_ptr())
        _lib.check(rc, "egx_vposer_encode")

    @torch.no_grad()
    def encode_mean(self, body_pose: torch.Tensor) -> torch.Tensor:
        if not body_pose.is_cuda:
            raise _lib.EgxError("VPoser encoder runs on the HIP device only (no CPU fallback)")
        x = body_pose.to(torch.float32).reshape(body_pose.shape[0], -1).contiguous()
        out = torch.empty(x.shape[0], 32, dtype=torch.float32, device=x.device)
        self.encode_mean_into(x, x.shape[1], x.shape[0], out)
        return out


PREDICTOR_CFG = {"body_repr": "ssm2_67", "h_dim": 256, "z_dim": 128, "t_his": 2, "t_pred": 18,
                 "use_drnn_mlp": True, "hdims_mlp": [512, 256], "residual": True}
REGRESSOR_CFG = {"body_repr": "ssm2_67", "h_dim": 128, "n_blocks": 10, "n_recur": 3, "actfun": "relu", "use_cont": True}
POLICY_CFG = {"h_dim": 512, "z_dim": 128, "n_blocks": 2, "n_recur": -1, "body_repr": "ssm2_67_condi_marker_map",
              "actfun": "lrelu", "is_stochastic": True, "min_logvar": -2.5, "max_logvar": 2.5, "reproj_factor": 0.5,
              "map_res": 16, "map_extent": 0.8}
