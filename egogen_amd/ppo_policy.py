"""GAMMAPPOPolicy: host-side mirror of crowd_ppo/ppo_policy.py (which subclasses tianshou.PPOPolicy) without
tianshou.  Same constructor arguments, same state_dict key set (incl. the duplicated `_actor_critic.*`
entries the tianshou parent creates, main_ppo.py:207-216 / SURVEY 8(b)), same loss and optimiser schedule.

Rollout-time pieces (policy forward, action sampling, critic values, GAE) run through libegogen_hip.so, and so does the
update: forward, loss and backward of a minibatch are one fixed launch chain (`egx_policy_train_step`, csrc/update3.hip) that
writes the gradients into one flat buffer, followed by the flat clip + AdamW kernels and an optional RCCL all-reduce of that
buffer for data parallelism (one process per GPU).  The autograd-node formulation (fused_ops.py, same kernels for its
products) stays as the checker of the chain and serves minibatches that are not a multiple of 32 rows.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from .models import ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, PolicyHipRunner

_EPS = float(np.finfo(np.float32).eps)  # tianshou BasePolicy._eps
_LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))
_UPDATE_PREC = {"f32": 0, "bf16x2": 2, "bf16": 1}


class RunningMeanStd:
    """tianshou.utils.RunningMeanStd [upstream]: running mean / variance of a data stream, merged batch by batch
    (Chan's parallel algorithm), as `BasePolicy.ret_rms` uses it for `reward_normalization` (ppo_policy.py:121-135)."""

    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, 0

    def update(self, x) -> None:
        x = np.asarray(x, np.float64).reshape(-1)
        self.update_from_moments(float(x.mean()), float(x.var()), x.size)

    def update_from_moments(self, bm: float, bv: float, bc) -> None:
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class RolloutBatch:
    """Time-major storage of one collect: n vector steps of A agents (+ the observation after the last step)."""

    def __init__(self, n: int, A: int, device):
        f = dict(dtype=torch.float32, device=device)
        self.n, self.A = n, A
        self.state = torch.zeros(n + 1, A, 2, 402, **f)
        self.ego = torch.zeros(n + 1, A, 2, 32, **f)
        self.dist = torch.zeros(n + 1, A, **f)
        self.time = torch.zeros(n + 1, A, **f)
        self.act = torch.zeros(n, A, 128, **f)
        self.mu = torch.zeros(n, A, 128, **f)
        self.logvar = torch.zeros(n, A, 128, **f)
        self.logp_old = torch.zeros(n, A, **f)
        self.rew = torch.zeros(n, A, **f)
        self.term = torch.zeros(n, A, dtype=torch.int32, device=device)
        self.values = torch.zeros(n + 1, A, **f)
        self.returns = torch.zeros(n, A, **f)
        self.adv = torch.zeros(n, A, **f)

    def store_obs(self, t: int, obs: Dict[str, torch.Tensor]):
        self.state[t].copy_(obs["state"])
        self.ego[t].copy_(obs["egosensing"])
        self.dist[t].copy_(obs["dist"].reshape(-1))
        self.time[t].copy_(obs["time"].reshape(-1))

    def obs_flat(self, upto: Optional[int] = None) -> Dict[str, torch.Tensor]:
        k = self.n if upto is None else upto
        return {"state": self.state[:k].reshape(-1, 2, 402), "egosensing": self.ego[:k].reshape(-1, 2, 32),
                "dist": self.dist[:k].reshape(-1), "time": self.time[:k].reshape(-1)}


class GAMMAPPOPolicy(nn.Module):
    def __init__(self, actor: GAMMAActor, critic: GAMMACritic, shared_net: GAMMAPolicyBase, optim: torch.optim.Optimizer,
                 dist_fn=None, discount_factor: float = 0.99, gae_lambda: float = 0.95, max_grad_norm: Optional[float] = None,
                 vf_coef: float = 0.5, ent_coef: float = 0.01, weight_kld: float = 1.0, reward_normalization: bool = False,
                 eps_clip: float = 0.2, dual_clip: Optional[float] = None, value_clip: bool = False,
                 advantage_normalization: bool = True, recompute_advantage: bool = False,
                 deterministic_eval: bool = False, max_batchsize: int = 256, seed: int = 0, **_ignored):
        super().__init__()
        # ppo_policy.py:80-86 / tianshou A2CPolicy: the options main_ppo.py exposes as --dual-clip / --value-clip / --rew-norm /
        # --recompute-adv (all off by default).  With any of the loss options on, the minibatch runs through the torch
        # expression of `minibatch_loss` (autograd nodes on the HIP products): the hand-written chain and the fused loss
        # kernel implement the default loss only.
        assert dual_clip is None or dual_clip > 1.0, "Dual-clip PPO parameter should greater than 1.0."
        self._dual_clip = dual_clip
        self._value_clip = bool(value_clip)
        self._rew_norm = bool(reward_normalization)
        self._recompute_adv = bool(recompute_advantage)
        self.ret_rms = RunningMeanStd()      # tianshou BasePolicy.ret_rms (returns' running variance, rew_norm only)
        self.actor, self.critic, self.shared_net = actor, critic, shared_net
        # tianshou's A2CPolicy builds ActorCritic(actor, critic); ppo_policy.py:90 is a bare annotation, so the
        # parent's object (WITHOUT shared_net) is what clip_grad_norm_ sees
        self._actor_critic = ActorCritic(actor, critic)
        self.optim = optim
        self._gamma, self._lambda = discount_factor, gae_lambda
        self._grad_norm = max_grad_norm
        self._weight_vf, self._weight_ent, self._weight_kld = vf_coef, ent_coef, weight_kld
        self._eps_clip = eps_clip
        self._norm_adv = bool(advantage_normalization)
        self._deterministic_eval = deterministic_eval
        self._batch = max_batchsize
        self._runner = PolicyHipRunner(shared_net, actor, critic)
        self._noise_gen: Optional[torch.Generator] = None
        self._seed = seed
        self._perm_gen = torch.Generator().manual_seed(seed)
        self._flat_grad: Optional[torch.Tensor] = None
        self._after_minibatch = None   # optional callable(i), called after optimiser step i of learn() (tests)
        self._layout = None
        self._graph_cache: dict = {}
        self.use_update_graph = bool(_ignored.get("use_update_graph", False))
        # data parallel + replayed graphs: ONE all-reduce of the whole flat gradient between the two graphs (default).
        # EGX_DP_OVERLAP=1 (opt-in): the chain in two halves, the actor + critic bucket all-reduced on a communication stream beside
        # the encoders' backward.  Opt-in because its stream ordering has only ever run with both ranks on one device over gloo
        # (the blocking collective serialises on the host there): no multi-GPU node has executed it, and a wrong bucket boundary
        # or a missed wait would corrupt gradients silently.  tests/test_multirank_gpu.py holds the two forms to each other and
        # checks that the heads half finalises exactly [0, n_clip) and the encoders half writes nothing below n_clip.
        self.overlap_allreduce = os.environ.get("EGX_DP_OVERLAP", "0") == "1"
        self.use_fused_loss = bool(_ignored.get("use_fused_loss", True))
        self._scale_cache = {}
        # the minibatch as a fixed chain of hand-written launches (csrc/update3.hip); EGX_TRAIN_STEP=0: the autograd nodes
        self.use_train_step = bool(_ignored.get("use_train_step", os.environ.get("EGX_TRAIN_STEP", "1") != "0"))
        self._train_handles: dict = {}
        # arithmetic of the chain's products (egx_policy_train_set_precision): "f32" = three bf16 terms per operand (2^-24),
        # "bf16x2" = two terms, "bf16" = operands rounded to bf16; fp32 accumulation and fp32 gradients in all of them
        self.update_precision = str(_ignored.get("update_precision", os.environ.get("EGX_UPDATE_PREC", "f32")))
        if self.update_precision not in _UPDATE_PREC:
            raise ValueError(f"update_precision must be one of {sorted(_UPDATE_PREC)}, got {self.update_precision!r}")
        # dense layers of the update as LinearFn nodes (library GEMMs + fused activation / bias-gradient / accumulation
        # kernels).  Their weight gradients are ACCUMULATED into the flat buffer: callers zero it once per minibatch.
        self.use_fused_linear = bool(_ignored.get("use_fused_linear", True))
        # clip + AdamW as two kernels over flat buffers (CUDA, single-group AdamW); see _flat_optimizer_ready
        self.use_flat_optimizer = bool(_ignored.get("use_flat_optimizer", os.environ.get("EGX_FLAT_OPTIMIZER", "1") != "0"))
        self._flat_opt_state = None
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # the data-parallel code path (global advantage moments, graph 1 | all-reduce | graph 2): taken with more than one
        # rank - and, for testing the RCCL calls between graph replays on a one-GPU box, with ONE rank when asked for
        self._dp = self.world_size > 1 or (os.environ.get("EGX_FORCE_DP_PATH") == "1" and dist.is_available() and dist.is_initialized())
        self.update_paths: dict = {}       # minibatches of learn() by formulation: "chain", "chain+graph", "autograd", "autograd+graph"
        self.allreduce_events: list = []   # [(start, stop)] torch.cuda.Event pairs, one consumed per gradient all-reduce
        self._allreduce_done: list = []

    # ---- rollout side (HIP) -------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, obs: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None, out: Optional[dict] = None):
        """ppo_policy.py:142-179.  Returns dict(act, mu (=z_mu), logvar (clamped z_logvar), logp, value)."""
        lib = _lib.load()
        out = self._runner.forward(obs, want_actor=True, want_critic=True, out=out)
        n = out["mu"].shape[0]
        dev = out["mu"].device
        deterministic = self._deterministic_eval and not self.training
        if not deterministic and noise is None:
            if self._noise_gen is None:
                self._noise_gen = torch.Generator(device=dev)
                # exploration noise differs between data-parallel ranks (parameters and the minibatch permutation do not)
                rank = dist.get_rank() if self.world_size > 1 else 0
                self._noise_gen.manual_seed(self._seed + 1 + 9973 * rank)
            noise = torch.randn(n, 128, generator=self._noise_gen, device=dev)
        if "act" not in out:
            out["act"] = torch.empty(n, 128, dtype=torch.float32, device=dev)
            out["logp"] = torch.empty(n, dtype=torch.float32, device=dev)
        rc = lib.egx_sample_action(_lib.ptr(out["mu"]), _lib.ptr(out["logvar"]), _lib.ptr(noise) if noise is not None else None,
                                   float(self.actor.min_logvar), float(self.actor.max_logvar), 1 if deterministic else 0, n,
                                   _lib.ptr(out["act"]), _lib.ptr(out["logp"]), _lib.current_stream_ptr())
        _lib.check(rc, "egx_sample_action")
        return out

    @torch.no_grad()
    def values(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """critic(shared_net(obs)) in chunks of max_batchsize rows is what the reference does (ppo_policy.py:110-112);
        one batched call computes the same rows."""
        return self._runner.forward(obs, want_actor=False, want_critic=True)["value"]

    @torch.no_grad()
    def process_fn(self, batch: RolloutBatch):
        """ppo_policy.py:93-140: values of obs / obs_next, GAE returns and advantages."""
        lib = _lib.load()
        n, A = batch.n, batch.A
        if getattr(batch, "rollout_values_valid", False) and not getattr(self, "_recomputing", False):
            # V(obs_0..n-1) were produced by the rollout forward passes with the same weights (the reference re-evaluates
            # them, ppo_policy.py:110-112, and gets the same numbers); only the observation after the last step is new
            last = {k: v[n * A:(n + 1) * A] for k, v in batch.obs_flat(n + 1).items()}
            batch.values[n].copy_(self.values(last))
        else:
            v = self.values(batch.obs_flat(n + 1))
            batch.values.copy_(v.reshape(n + 1, A))
        values = batch.values
        scale = 1.0
        if self._rew_norm:   # ppo_policy.py:121-123: the critic predicts NORMALISED returns: un-normalise v_s / v_s_ for the scan
            scale = float(np.sqrt(self.ret_rms.var + _EPS))
            values = (batch.values * scale).contiguous()
        rc = lib.egx_gae(_lib.ptr(values), _lib.ptr(batch.rew), _lib.ptr(batch.term), n, A, float(self._gamma),
                         float(self._lambda), _lib.ptr(batch.returns), _lib.ptr(batch.adv), _lib.current_stream_ptr())
        _lib.check(rc, "egx_gae")
        if self._rew_norm:   # :131-133 (one host round trip per collect, like the reference's numpy statistics)
            unnorm = batch.returns.detach().double()
            batch.returns.div_(scale)
            if self._dp:
                # one critic is shared by all ranks (gradients are all-reduced): its target scale must be the same everywhere, so
                # the running statistics take the moments of ALL ranks' returns (tianshou keeps them per process)
                mom = torch.stack([unnorm.sum(), (unnorm * unnorm).sum(), torch.tensor(float(unnorm.numel()), dtype=torch.float64,
                                                                                       device=unnorm.device)])
                dist.all_reduce(mom)
                s1, s2, cnt = (float(v) for v in mom.cpu())
                mean = s1 / cnt
                self.ret_rms.update_from_moments(mean, max(0.0, s2 / cnt - mean * mean), cnt)
            else:
                self.ret_rms.update(unnorm.cpu().numpy())
        return batch

    # ---- update side (autograd) ---------------------------------------------------------------------
    def _dist_params(self, obs):
        hx = self.shared_net(obs)
        (mu, logvar), _ = self.actor(hx)
        logvar = logvar.clamp(self.actor.min_logvar, self.actor.max_logvar)
        sigma = torch.exp(logvar) ** 0.5
        return hx, mu, sigma

    @staticmethod
    def log_prob(mu, sigma, act):
        return (-((act - mu) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - _LOG_SQRT_2PI).sum(-1)

    @staticmethod
    def entropy(sigma):
        return (0.5 + _LOG_SQRT_2PI + torch.log(sigma)).sum(-1)

    def _flat_layout(self):
        """(parameter, offset, numel) in optimiser order, every tensor starting on a 256-byte boundary (library GEMMs pick
        slower kernels for operands that are only 4-byte aligned); the padding elements stay zero in every flat buffer."""
        if self._layout is None:
            params = [p for g in self.optim.param_groups for p in g["params"]]
            off, lay = 0, []
            for p in params:
                lay.append((p, off, p.numel()))
                off += (p.numel() + 63) // 64 * 64
            self._layout, self._layout_total = lay, off
        return self._layout

    def _ensure_flat_grads(self):
        """One flat fp32 gradient buffer (13 168 001 floats + alignment padding) aliased by every .grad: a single
        in-place all-reduce."""
        lay = self._flat_layout()
        if self._flat_grad is None:
            self._flat_grad = torch.zeros(self._layout_total, dtype=torch.float32, device=lay[0][0].device)
        base = self._flat_grad.data_ptr()
        for p, off, n in lay:  # (re-)attach: zero_grad(set_to_none=True) or a foreign backward may have replaced a view
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self._flat_grad[off:off + n].view_as(p)

    def _loss_options(self) -> bool:
        return self._dual_clip is not None or self._value_clip

    def minibatch_loss(self, obs, act, adv, returns, logp_old, global_stats=None, v_s=None):
        """ppo_policy.py:189-241 for one minibatch.  With data parallelism `global_stats` = (mean, std, n_global):
        the advantage statistics and the loss normaliser of the GLOBAL minibatch."""
        n_local = adv.shape[0]
        if adv.is_cuda and self.use_fused_loss and not self._loss_options():
            return self._minibatch_loss_fused(obs, act, adv, returns, logp_old, global_stats)
        hx, mu, sigma = self._dist_params(obs)
        if self._norm_adv:
            if global_stats is None:
                mean, std = adv.mean(), adv.std()
            else:
                mean, std = global_stats[0], global_stats[1]
            adv = (adv - mean) / (std + _EPS)
        scale = 1.0 / n_local if global_stats is None else 1.0 / global_stats[2]
        lp = self.log_prob(mu, sigma, act)
        ratio = (lp - logp_old).exp().float()
        surr1 = ratio * adv
        surr2 = ratio.clamp(1.0 - self._eps_clip, 1.0 + self._eps_clip) * adv
        if self._dual_clip:      # ppo_policy.py:204-207
            clip1 = torch.min(surr1, surr2)
            clip2 = torch.max(clip1, self._dual_clip * adv)
            clip_loss = -torch.where(adv < 0, clip2, clip1).sum() * scale
        else:
            clip_loss = -torch.min(surr1, surr2).sum() * scale
        value = self.critic(hx).flatten()
        if self._value_clip:     # :216-221
            if v_s is None:
                raise ValueError("value_clip needs the old values v_s of the minibatch")
            v_clip = v_s + (value - v_s).clamp(-self._eps_clip, self._eps_clip)
            vf_loss = torch.max((returns - value).pow(2), (returns - v_clip).pow(2)).sum() * scale
        else:
            vf_loss = (returns - value).pow(2).sum() * scale
        ent_loss = self.entropy(sigma).sum() * scale
        kld_loss = 0.5 * mu.pow(2).sum() * scale / mu.shape[1]
        loss = clip_loss + self._weight_vf * vf_loss - self._weight_ent * ent_loss
        return loss, {"loss": loss, "loss/clip": clip_loss, "loss/vf": vf_loss, "loss/ent": ent_loss, "loss/kld": kld_loss,
                      "approx_kl": (logp_old - lp).sum() * scale}

    # ---- one optimiser step, eager or as replayed HIP graphs --------------------------------------------
    def _gather(self, batch: RolloutBatch, idx: torch.Tensor):
        N = batch.n * batch.A
        obs_all = batch.obs_flat()
        if idx.is_cuda and self.use_fused_loss and all(v.dtype == torch.float32 for v in obs_all.values()):
            from .fused_ops import gather_rows  # eight row gathers in one launch
            st, ego, di, ti, act, adv, ret, lpo = gather_rows(idx, [
                obs_all["state"].reshape(N, 804), obs_all["egosensing"].reshape(N, 64), obs_all["dist"].reshape(N, 1),
                obs_all["time"].reshape(N, 1), batch.act.reshape(N, 128), batch.adv.reshape(N, 1), batch.returns.reshape(N, 1),
                batch.logp_old.reshape(N, 1)])
            n = idx.shape[0]
            obs = {"state": st.reshape(n, 2, 402), "egosensing": ego.reshape(n, 2, 32), "dist": di.reshape(n), "time": ti.reshape(n)}
            return obs, act, adv.reshape(n), ret.reshape(n), lpo.reshape(n)
        obs = {k: v.index_select(0, idx) for k, v in obs_all.items()}
        return (obs, batch.act.reshape(N, 128).index_select(0, idx), batch.adv.reshape(N).index_select(0, idx),
                batch.returns.reshape(N).index_select(0, idx), batch.logp_old.reshape(N).index_select(0, idx))

    # ---- the minibatch as a fixed chain of hand-written launches (csrc/update3.hip) ---------------------------------
    def _train_handle(self, n: int):
        """`egx_policy_train` for minibatches of n rows, or None where the hand-written step does not apply (CPU tensors, a
        minibatch that is not a multiple of 32 rows, an optimiser the flat AdamW kernel does not cover, ablation switches)."""
        if not (self.use_train_step and self.use_fused_loss and self.use_fused_linear and n % 32 == 0 and self.actor.z_dim == 128
                and self._flat_opt_state == "ready" and self._flat_grad is not None and self._flat_grad.is_cuda and self._norm_adv
                and self._chain_shapes_ok() and not self._loss_options()):
            return None
        hs = self._train_handles.get(n)
        if hs is not None:
            return hs
        if self._graph_cache:
            # a captured (clip + AdamW + image refresh) graph re-makes the images of the handles that existed at capture time
            # only: a handle created later (the merged last minibatch of a pass has its own size) would keep forwarding
            # through weights that are k - 1 optimiser steps old.  Drop the graphs; the next minibatch re-captures them with
            # every handle in the refresh.
            self._graph_cache.clear()
        lib = _lib.load()
        self._ensure_flat_grads()
        w = self._runner._weights()
        g = _lib.PolicyGrads()
        s, a, c = self.shared_net, self.actor.pnet, self.critic.vnet
        gp = lambda t: t.grad.data_ptr()
        g.x_enc_w_ih, g.x_enc_w_hh, g.x_enc_b_ih, g.x_enc_b_hh = gp(s.x_enc.weight_ih_l0), gp(s.x_enc.weight_hh_l0), gp(s.x_enc.bias_ih_l0), gp(s.x_enc.bias_hh_l0)
        g.ego_enc_w_ih, g.ego_enc_w_hh, g.ego_enc_b_ih, g.ego_enc_b_hh = (gp(s.ego_enc.weight_ih_l0), gp(s.ego_enc.weight_hh_l0),
                                                                            gp(s.ego_enc.bias_ih_l0), gp(s.ego_enc.bias_hh_l0))
        for b in range(2):
            for k in range(2):
                g.actor_w[2 * b + k], g.actor_b[2 * b + k] = gp(a.layers[b].layers[k].weight), gp(a.layers[b].layers[k].bias)
                g.critic_w[2 * b + k], g.critic_b[2 * b + k] = gp(c.layers[b].layers[k].weight), gp(c.layers[b].layers[k].bias)
        g.actor_out_w, g.actor_out_b = gp(a.out_fc.weight), gp(a.out_fc.bias)
        g.critic_out_w, g.critic_out_b = gp(c.out_fc.weight), gp(c.out_fc.bias)
        h = C.c_void_p()
        _lib.check(lib.egx_policy_train_create(C.byref(w), C.byref(g), int(n), C.byref(h)), "egx_policy_train_create")
        _lib.check(lib.egx_policy_train_set_precision(h, _UPDATE_PREC[self.update_precision]), "egx_policy_train_set_precision")
        dev = self._flat_grad.device
        f = dict(dtype=torch.float32, device=dev)
        bufs = [torch.empty(n, 804, **f), torch.empty(n, 64, **f), torch.empty(n, 1, **f), torch.empty(n, 1, **f),
                torch.empty(n, 128, **f), torch.empty(n, 1, **f), torch.empty(n, 1, **f), torch.empty(n, 1, **f)]
        _lib.check(lib.egx_policy_train_bind(h, _lib.ptr(bufs[0]), _lib.ptr(bufs[1])), "egx_policy_train_bind")
        packed = _lib.PolicyPacked3()
        _lib.check(lib.egx_policy_train_packed(h, C.byref(packed)), "egx_policy_train_packed")
        hs = {"h": h, "bufs": bufs, "stats": torch.zeros(2, **f), "key": (w.x_enc_w_ih, g.x_enc_w_ih), "packed": packed}
        self._train_handles[n] = hs
        if len(self._train_handles) == 1:   # the rollout forward reads the images this handle keeps current
            self._runner.adopt_packed(packed, refresh=self._refresh_images)
            self._images_owner = n
        self._refresh_images()
        return hs

    def _chain_shapes_ok(self) -> bool:
        """csrc/update3.hip is written for the released policy: two 512-wide GRU encoders, 2 residual units of two 1152-wide
        layers with LeakyReLU(0.01), a 256-wide actor head and a scalar critic head.  Anything else trains through the
        autograd nodes."""
        ok = getattr(self, "_chain_ok", None)
        if ok is None:
            a, c, s = self.actor.pnet, self.critic.vnet, self.shared_net
            def block_ok(blk, nout):
                return (blk.residual and len(blk.layers) == 2 and blk.out_fc.out_features == nout and blk.out_fc.in_features == 1152
                        and all(m.act_name == "lrelu" and len(m.layers) == 2 and
                                all(fc.in_features == 1152 and fc.out_features == 1152 for fc in m.layers) for m in blk.layers))
            ok = (s.h_dim == 512 and s.x_enc.input_size == 402 and s.ego_enc.input_size == 32 and block_ok(a, 256) and block_ok(c, 1))
            self._chain_ok = ok
        return ok

    def _drop_train_handles(self):
        lib = _lib.load() if self._train_handles else None
        for hs in self._train_handles.values():
            lib.egx_policy_train_destroy(hs["h"])
        self._train_handles = {}
        self._runner._wstruct = None   # the rollout runner goes back to images of its own

    def __del__(self):
        try:
            self._drop_train_handles()
        except Exception:
            pass

    def _refresh_images(self):
        """Re-make the packed weight images of every update handle (one launch each) - after anything that changed the
        parameters."""
        if not self._train_handles:
            return
        lib, st = _lib.load(), _lib.current_stream_ptr()
        for hs in self._train_handles.values():
            _lib.check(lib.egx_policy_train_refresh(hs["h"], st), "egx_policy_train_refresh")

    def _fwd_bwd_train_step(self, hs, batch, idx, gstats, log_out, part: str = "all"):
        """gather + forward + loss + backward of one minibatch: 2 + ~21 launches, gradients written into the flat buffer.
        `part`: "all"; "heads" = everything up to the last actor / critic weight gradient (data-parallel training all-reduces that
        bucket while "encoders" - the GRU encoders' backward - runs)."""
        from .fused_ops import gather_rows
        lib, st = _lib.load(), _lib.current_stream_ptr()
        if part == "encoders":
            _lib.check(lib.egx_policy_train_step_encoders(hs["h"], st), "egx_policy_train_step_encoders")
            return
        N = batch.n * batch.A
        obs_all = batch.obs_flat()
        b = hs["bufs"]
        gather_rows(idx, [obs_all["state"].reshape(N, 804), obs_all["egosensing"].reshape(N, 64), obs_all["dist"].reshape(N, 1),
                          obs_all["time"].reshape(N, 1), batch.act.reshape(N, 128), batch.adv.reshape(N, 1), batch.returns.reshape(N, 1),
                          batch.logp_old.reshape(N, 1)], out=b)
        n = int(idx.shape[0])
        dev = b[0].device
        if gstats is None:
            _lib.check(lib.egx_adv_stats(_lib.ptr(b[5]), n, _lib.ptr(hs["stats"]), st), "egx_adv_stats")
            scale = self._scale_cache.get((n, dev))
            if scale is None:
                scale = torch.full((1,), 1.0 / n, dtype=torch.float32, device=dev)
                self._scale_cache[(n, dev)] = scale
        else:
            hs["stats"].copy_(torch.stack([gstats[0], gstats[1]]).float())
            scale = (1.0 / gstats[2]).reshape(1).float()
        fn = lib.egx_policy_train_step_heads if part == "heads" else lib.egx_policy_train_step
        rc = fn(hs["h"], _lib.ptr(b[2]), _lib.ptr(b[3]), _lib.ptr(b[4]), _lib.ptr(b[5]), _lib.ptr(b[6]), _lib.ptr(b[7]),
                _lib.ptr(hs["stats"]), _lib.ptr(scale), float(_EPS), float(self.actor.min_logvar),
                float(self.actor.max_logvar), float(self._eps_clip), float(self._weight_vf), float(self._weight_ent),
                _lib.ptr(log_out), st)
        _lib.check(rc, "egx_policy_train_step")

    def _fwd_bwd(self, batch, idx, gstats, log_out) -> str:
        """Forward, loss and backward of one minibatch into the flat gradient; returns which formulation ran ("chain": the
        hand-written launch chain of csrc/update3.hip, "autograd": the autograd nodes)."""
        hs = self._train_handle(int(idx.shape[0])) if idx.is_cuda else None
        if hs is not None:
            self._fwd_bwd_train_step(hs, batch, idx, gstats, log_out)
            return "chain"
        obs, act, adv, ret, lpo = self._gather(batch, idx)
        v_s = batch.values[:batch.n].reshape(-1).index_select(0, idx) if self._value_clip else None
        loss, terms = self.minibatch_loss(obs, act, adv, ret, lpo, gstats, v_s=v_s)
        self._flat_grad.zero_()
        loss.backward()
        if adv.is_cuda and self.use_fused_loss and self.use_fused_linear:
            # the ego encoder / critic branches of the fused forward ran on the side stream and autograd replays their
            # backward nodes there; what follows (all-reduce, clip, AdamW) reads the flat gradient on THIS stream
            from . import models as _m
            if _m._TWO_STREAM_UPDATE:
                torch.cuda.current_stream().wait_stream(_m._side_stream(adv.device))
        packed = terms.get("_packed")
        if packed is not None:  # the fused loss already holds the six terms in one tensor
            log_out.copy_(packed)
        else:
            log_out.copy_(torch.stack([terms[k].detach() for k in ("loss", "loss/clip", "loss/vf", "loss/ent", "loss/kld", "approx_kl")]))
        return "autograd"

    def _clip_and_step(self):
        if self.use_flat_optimizer and self._flat_opt_state == "ready":
            g = self.optim.param_groups[0]
            lib = _lib.load()
            b1, b2 = g["betas"]
            rc = lib.egx_adamw_clip_step(_lib.ptr(self._flat_p), _lib.ptr(self._flat_grad), _lib.ptr(self._flat_m), _lib.ptr(self._flat_v),
                                         self._flat_p.numel(), self._n_clip, float(self._grad_norm or 0.0), float(g["lr"]), float(b1),
                                         float(b2), float(g["eps"]), float(g["weight_decay"]), _lib.ptr(self._step_t),
                                         _lib.ptr(self._adamw_ws), _lib.current_stream_ptr())
            _lib.check(rc, "egx_adamw_clip_step")
            self._runner.mark_dirty()   # parameters written by address: the rollout runner's packed images are stale
            self._refresh_images()      # ... and so are the update's own (re-made by one launch, inside the captured graph too)
            return
        if self._grad_norm:
            nn.utils.clip_grad_norm_(self._actor_critic.parameters(), max_norm=self._grad_norm)
        self.optim.step()
        self._refresh_images()

    def _flat_optimizer_ready(self) -> bool:
        """Flat-buffer AdamW (egx_adamw_clip_step): parameters, exp_avg and exp_avg_sq of the single AdamW group are
        re-pointed to views of three flat buffers (values preserved), `optim.state` keeps its torch layout, so
        optim.state_dict() / load_state_dict() and the checkpoint format are unchanged.  Falls back to torch when the
        optimiser is not a plain single-group AdamW or the clipped parameters are not a prefix of its parameter list."""
        if self._flat_opt_state == "unsupported":
            return False
        opt = self.optim
        params = [p for g in opt.param_groups for p in g["params"]]
        if self._flat_opt_state is None:
            ok = (isinstance(opt, torch.optim.AdamW) and len(opt.param_groups) == 1 and not opt.param_groups[0].get("amsgrad", False)
                  and not opt.param_groups[0].get("maximize", False) and all(p.is_cuda and p.dtype == torch.float32 for p in params))
            clip = list(self._actor_critic.parameters())
            ok = ok and len(clip) <= len(params) and all(a is b for a, b in zip(clip, params))
            if not ok:
                self._flat_opt_state = "unsupported"
                return False
            dev = params[0].device
            lay = self._flat_layout()
            total = self._layout_total
            self._flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
            self._flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
            self._flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
            self._step_t = torch.zeros((), dtype=torch.float32, device=dev)
            self._adamw_ws = torch.zeros(int(_lib.load().egx_adamw_workspace_floats()), dtype=torch.float32, device=dev)
            # the clipped parameters are a prefix of the layout: clip length = start of the first unclipped tensor
            self._n_clip = lay[len(clip)][1] if len(clip) < len(lay) else total
            with torch.no_grad():
                for p, off, n in lay:
                    self._flat_p[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = self._flat_p[off:off + n].view_as(p)
            self._flat_opt_state = "ready"
            self._graph_cache.clear()  # parameter storage moved
            self._drop_train_handles()
        # (re-)adopt the optimiser state: first use, or optim.load_state_dict() replaced the tensors
        st0 = opt.state.get(params[0], {})
        if st0.get("exp_avg") is None or st0["exp_avg"].data_ptr() != self._flat_m.data_ptr():
            with torch.no_grad():
                for p, off, n in self._flat_layout():
                    st = opt.state[p]
                    mv, vv = self._flat_m[off:off + n].view_as(p), self._flat_v[off:off + n].view_as(p)
                    if torch.is_tensor(st.get("exp_avg")):
                        mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                    else:
                        mv.zero_(); vv.zero_()
                    if "step" in st and p is params[0]:
                        self._step_t.fill_(float(st["step"]))
                    st["exp_avg"], st["exp_avg_sq"], st["step"] = mv, vv, self._step_t
            self._graph_cache.clear()
        return True

    def optim_state_dict(self) -> dict:
        """`optim.state_dict()` in the stock torch.optim.AdamW layout the reference writes (main_ppo.py:207-216): with the
        flat optimiser every parameter's `step` is one shared device scalar and exp_avg / exp_avg_sq are views of flat
        buffers; a checkpoint must hold one CPU fp32 step tensor per parameter and independent moment tensors, or a stock
        AdamW that loads it would advance the shared step once per parameter."""
        sd = self.optim.state_dict()
        out_state = {}
        for k, st in sd["state"].items():
            ns = {}
            for name, v in st.items():
                if name == "step":
                    ns[name] = torch.tensor(float(v), dtype=torch.float32)
                elif torch.is_tensor(v):
                    ns[name] = v.detach().clone()
                else:
                    ns[name] = v
            out_state[k] = ns
        return {"state": out_state, "param_groups": sd["param_groups"]}

    def _global_adv_stats(self, batch: RolloutBatch, perm: torch.Tensor, spans) -> torch.Tensor:
        """[n_minibatches, 3] float32 = (mean, unbiased std, n) of every GLOBAL minibatch of one pass over the data
        (ppo_policy.py:192-195 evaluated on the concatenation of all ranks' rows): the float64 moments of all local
        minibatches go through ONE all-reduce per pass, nothing synchronises with the host."""
        adv = batch.adv.reshape(-1).double()
        sizes = {e - s for s, e in spans}
        if len(sizes) == 1:
            a = adv.index_select(0, perm[spans[0][0]:spans[-1][1]]).reshape(len(spans), -1)
            mom = torch.stack([a.sum(1), (a * a).sum(1), torch.full_like(a[:, 0], float(a.shape[1]))], dim=1)
        else:  # merge_last made the final minibatch longer
            rows = []
            for s, e in spans:
                a = adv.index_select(0, perm[s:e])
                rows.append(torch.stack([a.sum(), (a * a).sum(), torch.full_like(a[0], float(e - s))]))
            mom = torch.stack(rows)
        mom = mom.contiguous()
        dist.all_reduce(mom)
        ng = mom[:, 2]
        mean = mom[:, 0] / ng
        var = (mom[:, 1] - ng * mean * mean) / (ng - 1)            # unbiased, like Tensor.std()
        return torch.stack([mean, var.clamp(min=0).sqrt(), ng], dim=1).float()

    def _grad_buckets(self):
        """The flat gradient as two contiguous buckets: [0, n_clip) = actor + critic (what crowd_ppo clips; final when the heads'
        half of the chain is enqueued) and [n_clip, total) = the shared GRU encoders."""
        n = getattr(self, "_n_clip", None)
        g = self._flat_grad
        if n is None or n <= 0 or n >= g.numel():
            return [g]
        return [g[:n], g[n:]]

    def _all_reduce_grad(self, bucket=None):
        """Sum of the flat gradient (or one bucket of it) over the ranks, in place (RCCL over xGMI; every rank already scaled its
        loss by 1 / n_global, so the sum IS the gradient of the global minibatch mean) on the CURRENT stream.  Timed by bench.py
        through `allreduce_events` (events on the issuing stream)."""
        ev = self.allreduce_events.pop() if self.allreduce_events else None
        if ev is not None:
            ev[0].record()
        dist.all_reduce(self._flat_grad if bucket is None else bucket)
        if ev is not None:
            ev[1].record()
            self._allreduce_done.append(ev)

    def _comm_stream(self):
        s = getattr(self, "_comm_s", None)
        if s is None:
            s = self._comm_s = torch.cuda.Stream()
        return s

    def _graphs_for(self, batch: RolloutBatch, local_bs: int):
        """Capture (gather + forward + loss + backward) and (clip + AdamW) for a fixed minibatch size.  With one rank
        both halves live in one graph; with several, the RCCL all-reduces run eagerly between the two replays."""
        key = (id(batch), local_bs)
        g = self._graph_cache.get(key)
        if g is not None:
            return g
        dev = batch.act.device
        st = {"idx": torch.zeros(local_bs, dtype=torch.long, device=dev), "log": torch.zeros(6, device=dev),
              "gstats": torch.tensor([0.0, 1.0, float(local_bs * self.world_size)], device=dev), "use_gstats": self._dp}
        gs = (lambda: (st["gstats"][0], st["gstats"][1], st["gstats"][2])) if st["use_gstats"] else (lambda: None)
        try:
            # warm-up on a side stream (lazy library init, allocator pools) - it performs real optimiser steps on a
            # throw-away copy of the state, which is restored afterwards
            snap = {k: v.clone() for k, v in self.state_dict().items()}
            params = [p_ for g_ in self.optim.param_groups for p_ in g_["params"]]
            osnap = {id(p_): {k: v.clone() for k, v in self.optim.state.get(p_, {}).items() if torch.is_tensor(v)} for p_ in params}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._fwd_bwd(batch, st["idx"], gs(), st["log"])
                    self._clip_and_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g1 = torch.cuda.CUDAGraph()
            g2 = None
            if not self._dp:
                with torch.cuda.graph(g1):
                    st["path"] = self._fwd_bwd(batch, st["idx"], gs(), st["log"])
                    self._clip_and_step()
            else:
                hs = self._train_handle(local_bs) if st["idx"].is_cuda else None
                if hs is not None and len(self._grad_buckets()) == 2 and self.overlap_allreduce:
                    # the chain in two halves: after g1 (heads) the actor + critic bucket is final and its all-reduce runs on a
                    # side stream beside g1b (the encoders' backward)
                    with torch.cuda.graph(g1):
                        self._fwd_bwd_train_step(hs, batch, st["idx"], gs(), st["log"], part="heads")
                    g1b = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1b, pool=g1.pool()):
                        self._fwd_bwd_train_step(hs, batch, st["idx"], gs(), st["log"], part="encoders")
                    st["g1b"], st["path"] = g1b, "chain"
                else:
                    with torch.cuda.graph(g1):
                        st["path"] = self._fwd_bwd(batch, st["idx"], gs(), st["log"])
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self._clip_and_step()
            # capture does not execute, but the warm-up steps did: restore parameters and optimiser state IN PLACE
            # (the graphs hold the addresses of the live tensors)
            self.load_state_dict(snap)
            for p_ in params:
                for k, v in self.optim.state.get(p_, {}).items():
                    if torch.is_tensor(v):
                        old_v = osnap[id(p_)].get(k)
                        v.copy_(old_v) if old_v is not None else v.zero_()
            st["g1"], st["g2"] = g1, g2
            self._refresh_images()   # the packed weight images followed the warm-up steps: back to the restored parameters
        except Exception as e:  # capture unsupported for some op on this stack: run the same ops eagerly
            import warnings
            warnings.warn(f"PPO update graph capture failed ({type(e).__name__}: {e}); running the update eagerly")
            st["g1"] = st["g2"] = None
            st["failed"] = True
        self._graph_cache[key] = st
        return st

    def _minibatch_loss_fused(self, obs, act, adv, returns, logp_old, global_stats):
        """Same function as the torch expression above, with the loss and its gradients w.r.t. (mu, logvar, value)
        computed by one HIP kernel (egx_ppo_loss)."""
        from .fused_ops import PPOLossFn, PPOLossPackedFn
        zp = None
        if self.use_fused_linear:
            from .models import fused_update_forward
            self._ensure_flat_grads()
            if self.actor.z_dim == 128:   # the loss kernel reads the actor head's [mu | logvar] in place
                zp, value = fused_update_forward(self.shared_net, self.actor, self.critic, obs, packed=True)
            else:
                mu, logvar, value = fused_update_forward(self.shared_net, self.actor, self.critic, obs)
        else:
            hx = self.shared_net(obs)
            (mu, logvar), _ = self.actor(hx)
            value = self.critic(hx).flatten()
        n_local = adv.shape[0]
        dev = adv.device
        stats = None
        if self._norm_adv:
            if global_stats is None:
                from .fused_ops import adv_stats
                stats = adv_stats(adv)
            else:
                stats = torch.stack([global_stats[0], global_stats[1]]).float()
        if global_stats is None:
            scale = self._scale_cache.get((n_local, dev))
            if scale is None:  # constant: built once, outside any graph capture
                scale = torch.full((1,), 1.0 / n_local, dtype=torch.float32, device=dev)
                self._scale_cache[(n_local, dev)] = scale
        else:
            scale = (1.0 / global_stats[2]).reshape(1).float()
        if zp is not None:
            loss, terms = PPOLossPackedFn.apply(zp, value, act, adv, returns, logp_old, stats, scale, _EPS, self.actor.min_logvar,
                                                self.actor.max_logvar, self._eps_clip, self._weight_vf, self._weight_ent)
        else:
            loss, terms = PPOLossFn.apply(mu, logvar, value, act, adv, returns, logp_old, stats, scale, _EPS, self.actor.min_logvar,
                                          self.actor.max_logvar, self._eps_clip, self._weight_vf, self._weight_ent)
        return loss, {"loss": terms[0], "loss/clip": terms[1], "loss/vf": terms[2], "loss/ent": terms[3], "loss/kld": terms[4],
                      "approx_kl": terms[5], "_packed": terms}

    def learn(self, batch: RolloutBatch, batch_size: int, repeat: int) -> Dict[str, List[float]]:
        """ppo_policy.py:182-265.  `batch_size` is the GLOBAL minibatch size; each rank contributes batch_size/world."""
        self.train()
        self._ensure_flat_grads()
        if self.use_flat_optimizer and batch.act.is_cuda:
            self._flat_optimizer_ready()  # (re-)points parameters / optimiser state BEFORE anything is captured
        self._refresh_images()            # whatever changed the parameters since the last update (load_state_dict, ...)
        ws = self.world_size
        dp = self._dp
        N = batch.n * batch.A
        dev = batch.act.device
        local_bs = max(1, batch_size // ws)
        names = ("loss", "loss/clip", "loss/vf", "loss/ent", "loss/kld")
        stats = {k: [] for k in names}
        logs = []
        use_graph = self.use_update_graph and dev.type == "cuda"
        kld_rows = []
        for step in range(repeat):
            if self._recompute_adv and step > 0:    # ppo_policy.py:185-186: values of ALL observations with the current weights
                self._recomputing = True
                try:
                    # replayed update graphs write the parameters by address (no tensor version changes, and the
                    # mark_dirty() inside _clip_and_step ran at capture time only): the packed images the value pass
                    # reads must be re-made from the CURRENT weights first
                    self._runner.mark_dirty()
                    self._refresh_images()
                    self.process_fn(batch)
                finally:
                    self._recomputing = False
            if getattr(self, "_perm_queue", None):        # parity tests: the row order the reference's Batch.split drew
                perm = torch.as_tensor(self._perm_queue.pop(0), dtype=torch.long).to(dev)
            else:
                perm = torch.randperm(N, generator=self._perm_gen).to(dev)
            # Batch.split(size, shuffle=True, merge_last=True)
            bounds = list(range(0, N, local_bs))
            if len(bounds) > 1 and N - bounds[-1] < local_bs:
                bounds.pop()
            last_log = None
            spans = [(s, bounds[i + 1] if i + 1 < len(bounds) else N) for i, s in enumerate(bounds)]
            gstats_all = self._global_adv_stats(batch, perm, spans) if dp else None
            for i, (s, e) in enumerate(spans):
                idx = perm[s:e]
                st = self._graphs_for(batch, local_bs) if (use_graph and e - s == local_bs) else None
                if st is not None and st.get("g1") is not None:
                    st["idx"].copy_(idx)
                    if dp:
                        st["gstats"].copy_(gstats_all[i])
                    st["g1"].replay()
                    if dp:
                        if st.get("g1b") is not None:
                            # bucket 0 (actor + critic: 78 % of the bytes) is reduced on the communication stream while the
                            # encoders' backward runs on this one; bucket 1 follows it there; clip + AdamW wait for both
                            main, comm = torch.cuda.current_stream(), self._comm_stream()
                            b0, b1 = self._grad_buckets()
                            comm.wait_stream(main)
                            with torch.cuda.stream(comm):
                                self._all_reduce_grad(b0)
                            st["g1b"].replay()
                            comm.wait_stream(main)
                            with torch.cuda.stream(comm):
                                self._all_reduce_grad(b1)
                            main.wait_stream(comm)
                        else:
                            self._all_reduce_grad()
                        st["g2"].replay()
                    last_log = st["log"].clone()
                    self.update_paths[st["path"] + "+graph"] = self.update_paths.get(st["path"] + "+graph", 0) + 1
                else:
                    gstats = None
                    if dp:
                        gstats = (gstats_all[i, 0], gstats_all[i, 1], gstats_all[i, 2])
                    log = torch.zeros(6, device=dev)
                    path = self._fwd_bwd(batch, idx, gstats, log)
                    if dp:
                        self._all_reduce_grad()
                    self._clip_and_step()
                    last_log = log
                    self.update_paths[path] = self.update_paths.get(path, 0) + 1
                logs.append(last_log[:5])
                if self._after_minibatch is not None:   # parity tests: the flat gradient / clip coefficient of THIS optimiser step
                    self._after_minibatch(len(logs) - 1)
                # `loss/kld` of ppo_policy.py:232 reads minibatch.z_mu - the means the ROLLOUT stored - not the current network's:
                # 0.5 mean(mu_rollout^2) over the minibatch's rows, evaluated for all minibatches at once after the loop
                kld_rows.append(idx)
            # early stop on the last minibatch's approximate KL (ppo_policy.py:252-257); inert at repeat=1
            if repeat > 1 and last_log is not None:
                kl = last_log[5].clone()
                if dp:
                    dist.all_reduce(kl)
                if float(kl.item()) >= 0.02:
                    break
        self._runner.mark_dirty()   # replayed graphs update the parameters by address: re-pack before the next rollout forward
        # a captured refresh writes the planes of the precision in force at capture time; one eager refresh here writes those of
        # the precision in force NOW (e.g. egx_policy_set_precision(0) for an fp32 evaluation after bf16x2 training)
        self._refresh_images()
        if logs:
            L = torch.stack(logs)
            mu2 = batch.mu.reshape(N, -1).pow(2).mean(1)               # [N]: per-row mean of the rollout's mu^2
            if len({int(r.shape[0]) for r in kld_rows}) == 1:
                kld = 0.5 * mu2.index_select(0, torch.cat(kld_rows)).reshape(len(kld_rows), -1).mean(1)
            else:
                kld = torch.stack([0.5 * mu2.index_select(0, r).mean() for r in kld_rows])
            if dp:
                kld = kld / ws                                          # summed over the ranks below: the global minibatch mean
            L[:, 4] = kld
            if dp:
                dist.all_reduce(L)
            for row in L.cpu().tolist():
                for k, v in zip(names, row):
                    stats[k].append(v)
        return stats
