"""Custom autograd nodes backed by libegogen_hip.so: GRU gate math (forward + backward), the fused clipped-PPO loss with
its gradients, and the dense layer `LinearFn` / `ResMLPFn` / `GRUSeqFn` nodes of the training operators (marker predictor,
body regressor, and the checker path of the PPO update).  Every matrix product of these nodes - x W^T + b forward, g^T x
and g W backward - is `egx_gemm3`: operands packed as bf16 triples, six partial products on the bf16 matrix pipe, fp32
accumulation (fp32-equivalent; the dense kernel of the rollout and of the update chain).  Bias, activation and residual run
in that kernel's epilogue, weight gradients accumulate straight into the flat gradient buffer (`res` aliasing `out`): no
library GEMM and no per-parameter AccumulateGrad kernels."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_GEMM_WS = {}


def _workspace(dev, nbytes):
    """Operand images of one product; one buffer per (device, stream), since products on different streams may overlap."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)   # owned by the graph's pool
    key = (str(dev), int(torch.cuda.current_stream(dev).cuda_stream))
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=dev)
        _GEMM_WS[key] = ws
    return ws


def gemm3(A, trans_a, B, trans_b, bias=None, act=0, slope=0.0, res=None, out=None, out_act=None):
    """out[M,N] = act(op(A) op(B)^T + bias) + res on 2-D fp32 tensors with unit inner stride (`egx_gemm3`).  op(A) is [M,K]:
    A itself, or A^T when trans_a (A stored [K,M]); op(B) is [N,K] likewise.  `res` may be `out` (accumulate in place)."""
    lib = _lib.load()
    for t in (A, B):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
            raise _lib.EgxError("gemm3 operands must be 2-D fp32 device tensors with unit inner stride")
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N, K2 = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != K2:
        raise ValueError(f"gemm3: reduction lengths differ ({K} vs {K2})")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    nbytes = int(lib.egx_gemm3_workspace_bytes(M, N, K))
    ws = _workspace(A.device, nbytes)
    for t in (res, out, out_act):
        if t is not None and (t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or tuple(t.shape) != (M, N)):
            raise _lib.EgxError("gemm3: res / out / out_act must be [M,N] fp32 with unit inner stride")
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())     # row strides are passed explicitly
    _lib.check(lib.egx_gemm3(p(A), A.stride(0), int(bool(trans_a)), p(B), B.stride(0), int(bool(trans_b)), M, N, K,
                             _lib.ptr(bias) if bias is not None else None, int(act), float(slope),
                             p(res), res.stride(0) if res is not None else 0, p(out), out.stride(0), p(out_act),
                             out_act.stride(0) if out_act is not None else 0, _lib.ptr(ws), nbytes, _lib.current_stream_ptr()),
               "egx_gemm3")
    return out


class GRUPointwiseFn(torch.autograd.Function):
    """h = (1-z) n + z h_prev with r,z,n from gi = x W_ih^T + b_ih, gh = h_prev W_hh^T + b_hh (gate order r,z,n)."""

    @staticmethod
    def forward(ctx, gi, gh, hprev):
        lib = _lib.load()
        gi, gh, hprev = gi.contiguous(), gh.contiguous(), hprev.contiguous()
        M, H = hprev.shape
        h = torch.empty_like(hprev)
        _lib.check(lib.egx_gru_pointwise(_lib.ptr(gi), _lib.ptr(gh), _lib.ptr(hprev), H, _lib.ptr(h), H, M, H,
                                         _lib.current_stream_ptr()), "egx_gru_pointwise")
        ctx.save_for_backward(gi, gh, hprev)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        gi, gh, hprev = ctx.saved_tensors
        M, H = hprev.shape
        dh = dh.contiguous()
        dgi, dgh, dhp = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(hprev)
        _lib.check(lib.egx_gru_pointwise_bwd(_lib.ptr(gi), _lib.ptr(gh), _lib.ptr(hprev), _lib.ptr(dh), M, H, _lib.ptr(dgi),
                                             _lib.ptr(dgh), _lib.ptr(dhp), _lib.current_stream_ptr()), "egx_gru_pointwise_bwd")
        return dgi, dgh, dhp


class PPOLossFn(torch.autograd.Function):
    """loss, terms[6] = egx_ppo_loss(...); gradients w.r.t. mu, raw logvar and value are produced by the same kernel."""

    @staticmethod
    def forward(ctx, mu, logvar, value, act, adv, ret, logp_old, adv_stats, scale, adv_eps, min_lv, max_lv, eps_clip, vf_coef,
                ent_coef):
        lib = _lib.load()
        mu, logvar, value = mu.contiguous(), logvar.contiguous(), value.contiguous().reshape(-1)
        n = mu.shape[0]
        g_mu, g_lv, g_v = torch.empty_like(mu), torch.empty_like(logvar), torch.empty_like(value)
        terms = torch.empty(6, dtype=torch.float32, device=mu.device)
        rc = lib.egx_ppo_loss(_lib.ptr(mu), _lib.ptr(logvar), _lib.ptr(value), _lib.ptr(act.contiguous()), _lib.ptr(adv.contiguous()),
                              _lib.ptr(ret.contiguous()), _lib.ptr(logp_old.contiguous()),
                              _lib.ptr(adv_stats) if adv_stats is not None else None, _lib.ptr(scale), float(adv_eps),
                              float(min_lv), float(max_lv), float(eps_clip), float(vf_coef), float(ent_coef), n,
                              _lib.ptr(g_mu), _lib.ptr(g_lv), _lib.ptr(g_v), _lib.ptr(terms), _lib.current_stream_ptr())
        _lib.check(rc, "egx_ppo_loss")
        ctx.save_for_backward(g_mu, g_lv, g_v)
        ctx.value_shape = value.shape
        ctx.mark_non_differentiable(terms)
        return terms[0].clone(), terms

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        g_mu, g_lv, g_v = ctx.saved_tensors
        return (grad_loss * g_mu, grad_loss * g_lv, (grad_loss * g_v).reshape(ctx.value_shape)) + (None,) * 12


class PPOLossPackedFn(torch.autograd.Function):
    """PPOLossFn on the actor head's raw output zp[n,256] = [mu | logvar]: the kernel reads both halves in place and writes one
    gradient tensor, so neither the split nor the re-join of the two halves costs a copy / fill / add."""

    @staticmethod
    def forward(ctx, zp, value, act, adv, ret, logp_old, adv_stats, scale, adv_eps, min_lv, max_lv, eps_clip, vf_coef, ent_coef):
        lib = _lib.load()
        zp, value = zp.contiguous(), value.contiguous().reshape(-1)
        n = zp.shape[0]
        g_zp, g_v = torch.empty_like(zp), torch.empty_like(value)
        terms = torch.empty(6, dtype=torch.float32, device=zp.device)
        rc = lib.egx_ppo_loss_packed(_lib.ptr(zp), _lib.ptr(value), _lib.ptr(act.contiguous()), _lib.ptr(adv.contiguous()),
                                     _lib.ptr(ret.contiguous()), _lib.ptr(logp_old.contiguous()),
                                     _lib.ptr(adv_stats) if adv_stats is not None else None, _lib.ptr(scale), float(adv_eps),
                                     float(min_lv), float(max_lv), float(eps_clip), float(vf_coef), float(ent_coef), n,
                                     _lib.ptr(g_zp), _lib.ptr(g_v), _lib.ptr(terms), _lib.current_stream_ptr())
        _lib.check(rc, "egx_ppo_loss_packed")
        ctx.save_for_backward(g_zp, g_v)
        ctx.value_shape = value.shape
        ctx.mark_non_differentiable(terms)
        return terms[0].clone(), terms

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        g_zp, g_v = ctx.saved_tensors
        return (grad_loss * g_zp, (grad_loss * g_v).reshape(ctx.value_shape)) + (None,) * 12


def posenc_dist_time(dist: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
    """[n] , [n] -> [n,128] = [posenc(dist) | posenc(time)] (no gradient: both are observations)."""
    lib = _lib.load()
    n = dist.shape[0]
    out = torch.empty(n, 128, dtype=torch.float32, device=dist.device)
    _lib.check(lib.egx_posenc(_lib.ptr(dist.contiguous()), _lib.ptr(time.contiguous()), n, _lib.ptr(out), _lib.current_stream_ptr()),
               "egx_posenc")
    return out


ACT_CODE = {None: 0, "none": 0, "tanh": 1, "relu": 2, "lrelu": 3, "leaky_relu": 3}


class LinearFn(torch.autograd.Function):
    """out = act(x W^T + b) (+ res).  The gradients of W and b are ACCUMULATED into `wg` / `bg` (views of the flat
    gradient buffer, zeroed once per minibatch by the caller) inside backward, and None is returned for them, so autograd
    launches no per-parameter accumulation kernels.  nn.Linear + activation of models_policy_ppo.py:24-39."""

    @staticmethod
    def forward(ctx, x, W, b, wg, bg, act, slope, res):
        x = x.contiguous()
        if res is not None:
            res = res.contiguous()
        M, N = x.shape[0], W.shape[0]
        # the activation BEFORE the residual is what backward needs; without a residual it is the output itself
        a = torch.empty(M, N, dtype=torch.float32, device=x.device) if (act != 0 and res is not None) else None
        out = gemm3(x, False, W, False, bias=b, act=act, slope=slope, res=res, out_act=a)
        ctx.save_for_backward(x, W, (a if a is not None else out) if act != 0 else None)
        ctx.wg, ctx.bg, ctx.act, ctx.slope, ctx.has_res = wg, bg, int(act), float(slope), res is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, W, a = ctx.saved_tensors
        dout = dout.contiguous()
        M, N = dout.shape
        g = torch.empty_like(dout) if ctx.act != 0 else dout
        _lib.check(lib.egx_act_bwd_colsum(_lib.ptr(dout), _lib.ptr(a) if a is not None else None,
                                          _lib.ptr(g) if ctx.act != 0 else None, _lib.ptr(ctx.bg), M, N, ctx.act, ctx.slope,
                                          _lib.current_stream_ptr()), "egx_act_bwd_colsum")
        gemm3(g, True, x, True, res=ctx.wg, out=ctx.wg)                       # wg += g^T x
        dx = gemm3(g, False, W, True) if ctx.needs_input_grad[0] else None    # g W
        return dx, None, None, None, None, None, None, (dout if ctx.has_res else None)


class ResMLPFn(torch.autograd.Function):
    """out = act(act(x W1^T + b1) W2^T + b2) + x: one residual unit of MLPBlock (models_policy_ppo.py:233-274) as ONE node.
    The same kernels and products as two LinearFn nodes, but the two gradient paths into x (through W1 and through the
    skip connection) are summed by the last product (the incoming gradient as its residual operand) instead of by an extra
    element-wise kernel of the autograd engine."""

    @staticmethod
    def forward(ctx, x, W1, b1, wg1, bg1, W2, b2, wg2, bg2, act, slope):
        x = x.contiguous()
        a1 = gemm3(x, False, W1, False, bias=b1, act=act, slope=slope)
        a2 = torch.empty(x.shape[0], W2.shape[0], dtype=torch.float32, device=x.device)
        out = gemm3(a1, False, W2, False, bias=b2, act=act, slope=slope, res=x, out_act=a2)
        ctx.save_for_backward(x, W1, a1, W2, a2)
        ctx.g = (wg1, bg1, wg2, bg2)
        ctx.act, ctx.slope = int(act), float(slope)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib, st = _lib.load(), _lib.current_stream_ptr()
        x, W1, a1, W2, a2 = ctx.saved_tensors
        wg1, bg1, wg2, bg2 = ctx.g
        dout = dout.contiguous()
        M = dout.shape[0]
        g2 = torch.empty_like(dout)
        _lib.check(lib.egx_act_bwd_colsum(_lib.ptr(dout), _lib.ptr(a2), _lib.ptr(g2), _lib.ptr(bg2), M, g2.shape[1], ctx.act, ctx.slope, st),
                   "egx_act_bwd_colsum")
        gemm3(g2, True, a1, True, res=wg2, out=wg2)
        d1 = gemm3(g2, False, W2, True)
        g1 = torch.empty_like(d1)
        _lib.check(lib.egx_act_bwd_colsum(_lib.ptr(d1), _lib.ptr(a1), _lib.ptr(g1), _lib.ptr(bg1), M, g1.shape[1], ctx.act, ctx.slope, st),
                   "egx_act_bwd_colsum")
        gemm3(g1, True, x, True, res=wg1, out=wg1)
        dx = gemm3(g1, False, W1, True, res=dout) if ctx.needs_input_grad[0] else None
        return (dx,) + (None,) * 10


def res_mlp(x, fc1: torch.nn.Linear, fc2: torch.nn.Linear, act="relu", slope=0.01):
    for fc in (fc1, fc2):
        if fc.weight.grad is None or fc.bias.grad is None:
            raise _lib.EgxError("ResMLPFn needs pre-allocated gradient views (GAMMAPPOPolicy._ensure_flat_grads)")
    if ACT_CODE[act] == 0:
        raise ValueError("res_mlp: the unit has an activation after both layers")
    return ResMLPFn.apply(x, fc1.weight, fc1.bias, fc1.weight.grad, fc1.bias.grad, fc2.weight, fc2.bias, fc2.weight.grad, fc2.bias.grad,
                          ACT_CODE[act], slope)


def linear_fn(x, weight, bias, act="none", slope=0.01, res=None):
    if weight.grad is None or bias.grad is None:
        raise _lib.EgxError("LinearFn needs pre-allocated gradient views (GAMMAPPOPolicy._ensure_flat_grads)")
    return LinearFn.apply(x, weight, bias, weight.grad, bias.grad, ACT_CODE[act], slope, res)


def linear_act(x, lin: torch.nn.Linear, act="none", slope=0.01, res=None):
    """LinearFn on an nn.Linear whose .grad tensors are views of the flat gradient buffer."""
    if lin.weight.grad is None or lin.bias.grad is None:
        raise _lib.EgxError("linear_act needs pre-allocated gradient views (GAMMAPPOPolicy._ensure_flat_grads)")
    return LinearFn.apply(x, lin.weight, lin.bias, lin.weight.grad, lin.bias.grad, ACT_CODE[act], slope, res)


def gather_rows(idx: torch.Tensor, srcs, out=None):
    """dst[t][r] = src[t][idx[r]] for up to 8 dense fp32 tensors (leading dimension = rows) in ONE launch."""
    import ctypes as C
    lib = _lib.load()
    n = int(idx.shape[0])
    k = len(srcs)
    srcs = [s if s.is_contiguous() else s.contiguous() for s in srcs]
    if out is None:
        out = [torch.empty((n,) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device) for s in srcs]
    widths = [int(s[0].numel()) for s in srcs]
    sp = (C.c_void_p * k)(*[s.data_ptr() for s in srcs])
    dp = (C.c_void_p * k)(*[o.data_ptr() for o in out])
    wp = (C.c_int * k)(*widths)
    _lib.check(lib.egx_gather_rows(_lib.ptr(idx), n, k, sp, wp, dp, _lib.current_stream_ptr()), "egx_gather_rows")
    return out


def adv_stats(adv: torch.Tensor) -> torch.Tensor:
    """[mean, unbiased std] of a minibatch of advantages, one launch."""
    lib = _lib.load()
    out = torch.empty(2, dtype=torch.float32, device=adv.device)
    _lib.check(lib.egx_adv_stats(_lib.ptr(adv.contiguous()), int(adv.numel()), _lib.ptr(out), _lib.current_stream_ptr()),
               "egx_adv_stats")
    return out


class GRUSeqFn(torch.autograd.Function):
    """Last hidden state of a one-layer nn.GRU over x[T*nb, in] (time-major, zero initial state) as ONE autograd node:
    the input products of all steps are one product, each step is a recurrent product + the fused gate kernel, and backward
    writes the gate gradients of every step straight into one [T*nb, 3H] buffer (no slicing / expand / accumulate nodes).
    Parameter gradients are accumulated into the flat-buffer views like LinearFn; x gets no gradient (observations)."""

    @staticmethod
    def forward(ctx, x2, w_ih, b_ih, w_hh, b_hh, g_w_ih, g_b_ih, g_w_hh, g_b_hh, T):
        lib = _lib.load()
        x2 = x2.contiguous()
        nb = x2.shape[0] // T
        H = w_hh.shape[1]
        st = _lib.current_stream_ptr()
        gi = gemm3(x2, False, w_ih, False, bias=b_ih)
        gh = torch.empty(T, nb, 3 * H, dtype=torch.float32, device=x2.device)
        hs = torch.zeros(T + 1, nb, H, dtype=torch.float32, device=x2.device)  # hs[0] = initial state
        gh[0].copy_(b_hh.unsqueeze(0).expand(nb, 3 * H))
        for t in range(T):
            if t > 0:
                gemm3(hs[t], False, w_hh, False, bias=b_hh, out=gh[t])
            _lib.check(lib.egx_gru_pointwise(_lib.ptr(gi[t * nb:(t + 1) * nb]), _lib.ptr(gh[t]), _lib.ptr(hs[t]), H, _lib.ptr(hs[t + 1]), H,
                                             nb, H, st), "egx_gru_pointwise")
        ctx.save_for_backward(x2, w_hh, gi, gh, hs)
        ctx.views = (g_w_ih, g_b_ih, g_w_hh, g_b_hh)
        ctx.T = T
        return hs[T]

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        x2, w_hh, gi, gh, hs = ctx.saved_tensors
        g_w_ih, g_b_ih, g_w_hh, g_b_hh = ctx.views
        T = ctx.T
        nb, H = hs.shape[1], hs.shape[2]
        st = _lib.current_stream_ptr()
        dgi = torch.empty_like(gi)
        dgh = torch.empty_like(gh)
        dhp = torch.empty(nb, H, dtype=torch.float32, device=dh.device)
        dh = dh.contiguous()
        for t in range(T - 1, -1, -1):
            _lib.check(lib.egx_gru_pointwise_bwd(_lib.ptr(gi[t * nb:(t + 1) * nb]), _lib.ptr(gh[t]), _lib.ptr(hs[t]), _lib.ptr(dh), nb, H,
                                                 _lib.ptr(dgi[t * nb:(t + 1) * nb]), _lib.ptr(dgh[t]), _lib.ptr(dhp), st),
                       "egx_gru_pointwise_bwd")
            if t > 0:  # through gh_t = h_t W_hh^T + b_hh
                dh = gemm3(dgh[t], False, w_hh, True, res=dhp)
                dhp = torch.empty_like(dhp)
        dgh2 = dgh.reshape(T * nb, 3 * H)
        _lib.check(lib.egx_act_bwd_colsum(_lib.ptr(dgh2), None, None, _lib.ptr(g_b_hh), T * nb, 3 * H, 0, 0.0, st), "egx_act_bwd_colsum")
        _lib.check(lib.egx_act_bwd_colsum(_lib.ptr(dgi), None, None, _lib.ptr(g_b_ih), T * nb, 3 * H, 0, 0.0, st), "egx_act_bwd_colsum")
        if T > 1:
            gemm3(dgh2[nb:], True, hs[1:T].reshape((T - 1) * nb, H), True, res=g_w_hh, out=g_w_hh)
        gemm3(dgi, True, x2, True, res=g_w_ih, out=g_w_ih)
        return (None,) * 10


class FlatGrads:
    """One flat fp32 gradient buffer aliased by the `.grad` of every parameter of `module` (256-byte aligned slices): what
    LinearFn / GRUSeqFn accumulate into.  `zero()` before every backward, `attach()` after anything that may have replaced
    a `.grad` (zero_grad(set_to_none=True), load_state_dict into new tensors)."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters()]
        off, self.layout = 0, []
        for p in self.params:
            self.layout.append((p, off, p.numel()))
            off += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.params[0].device)
        self.attach()

    def attach(self):
        base = self.flat.data_ptr()
        for p, off, n in self.layout:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self.flat[off:off + n].view_as(p)

    def zero(self):
        self.attach()
        self.flat.zero_()
