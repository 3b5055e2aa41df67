"""GAMMAPPOPolicy: host-side mirror of crowd_ppo/ppo_policy.py (which subclasses tianshou.PPOPolicy) without
tianshou.  Same constructor arguments, same state_dict key set (incl. the duplicated `_actor_critic.*`
entries the tianshou parent creates, main_ppo.py:207-216 / SURVEY 8(b)), same loss and optimiser schedule.

Rollout-time pieces (policy forward, action sampling, critic values, GAE) run through libegogen_hip.so; the
update uses torch autograd on the same parameter storage (rocBLAS GEMMs), with an optional RCCL all-reduce
of one flat gradient buffer for data parallelism (one process per GPU).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from .models import ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, PolicyHipRunner

_EPS = float(np.finfo(np.float32).eps)  # tianshou BasePolicy._eps
_LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


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
        if dual_clip is not None or value_clip or reward_normalization or recompute_advantage:
            raise NotImplementedError("main_ppo.py runs with dual_clip=None, value_clip=0, rew_norm=False, recompute_adv=0")
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
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

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
                self._noise_gen.manual_seed(self._seed + 1)
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
        v = self.values(batch.obs_flat(n + 1))
        batch.values.copy_(v.reshape(n + 1, A))
        rc = lib.egx_gae(_lib.ptr(batch.values), _lib.ptr(batch.rew), _lib.ptr(batch.term), n, A, float(self._gamma),
                         float(self._lambda), _lib.ptr(batch.returns), _lib.ptr(batch.adv), _lib.current_stream_ptr())
        _lib.check(rc, "egx_gae")
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

    def _ensure_flat_grads(self):
        """One flat fp32 gradient buffer (13 168 001 floats) aliased by every .grad: a single in-place all-reduce."""
        if self._flat_grad is not None:
            return
        params = [p for g in self.optim.param_groups for p in g["params"]]
        total = sum(p.numel() for p in params)
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        off = 0
        for p in params:
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()

    def minibatch_loss(self, obs, act, adv, returns, logp_old, global_stats=None):
        """ppo_policy.py:189-241 for one minibatch.  With data parallelism `global_stats` = (mean, std, n_global):
        the advantage statistics and the loss normaliser of the GLOBAL minibatch."""
        hx, mu, sigma = self._dist_params(obs)
        n_local = adv.shape[0]
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
        clip_loss = -torch.min(surr1, surr2).sum() * scale
        value = self.critic(hx).flatten()
        vf_loss = (returns - value).pow(2).sum() * scale
        ent_loss = self.entropy(sigma).sum() * scale
        kld_loss = 0.5 * mu.pow(2).sum() * scale / mu.shape[1]
        loss = clip_loss + self._weight_vf * vf_loss - self._weight_ent * ent_loss
        return loss, {"loss": loss, "loss/clip": clip_loss, "loss/vf": vf_loss, "loss/ent": ent_loss, "loss/kld": kld_loss,
                      "approx_kl": (logp_old - lp).sum() * scale}

    def learn(self, batch: RolloutBatch, batch_size: int, repeat: int) -> Dict[str, List[float]]:
        """ppo_policy.py:182-265.  `batch_size` is the GLOBAL minibatch size; each rank contributes batch_size/world."""
        self.train()
        self._ensure_flat_grads()
        ws = self.world_size
        N = batch.n * batch.A
        local_bs = max(1, batch_size // ws)
        obs_all = batch.obs_flat()
        act_all = batch.act.reshape(N, 128)
        adv_all, ret_all, lpo_all = batch.adv.reshape(N), batch.returns.reshape(N), batch.logp_old.reshape(N)
        stats = {k: [] for k in ("loss", "loss/clip", "loss/vf", "loss/ent", "loss/kld")}
        logs = []
        for _ in range(repeat):
            perm = torch.randperm(N, generator=self._perm_gen).to(act_all.device)
            # Batch.split(size, shuffle=True, merge_last=True)
            bounds = list(range(0, N, local_bs))
            if len(bounds) > 1 and N - bounds[-1] < local_bs:
                bounds.pop()
            kl = None
            for i, s in enumerate(bounds):
                e = bounds[i + 1] if i + 1 < len(bounds) else N
                idx = perm[s:e]
                obs = {k: v[idx] for k, v in obs_all.items()}
                adv = adv_all[idx]
                gstats = None
                if ws > 1:
                    mom = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), device=adv.device)]).double()
                    dist.all_reduce(mom)
                    ng = mom[2]
                    mean = mom[0] / ng
                    var = (mom[1] - ng * mean * mean) / (ng - 1)       # unbiased, like Tensor.std()
                    gstats = (mean.float(), var.clamp(min=0).sqrt().float(), ng.float())
                loss, terms = self.minibatch_loss(obs, act_all[idx], adv, ret_all[idx], lpo_all[idx], gstats)
                self._flat_grad.zero_()
                loss.backward()
                if ws > 1:
                    dist.all_reduce(self._flat_grad)
                if self._grad_norm:
                    nn.utils.clip_grad_norm_(self._actor_critic.parameters(), max_norm=self._grad_norm)
                self.optim.step()
                logs.append(torch.stack([terms[k].detach() for k in stats]))
                kl = terms["approx_kl"].detach()
            # early stop on the last minibatch's approximate KL (ppo_policy.py:252-257); inert at repeat=1
            if repeat > 1 and kl is not None:
                if ws > 1:
                    dist.all_reduce(kl)
                if float(kl.item()) >= 0.02:
                    break
        if logs:
            L = torch.stack(logs)
            if ws > 1:
                dist.all_reduce(L)
            L = L.cpu().tolist()
            for row in L:
                for k, v in zip(stats, row):
                    stats[k].append(v)
        return stats
