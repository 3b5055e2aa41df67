"""PPO pieces of crowd_ppo/ppo_policy.py, CPU restatement.  TEST INFRASTRUCTURE ONLY.

  gae_returns   ppo_policy.py:105-140 -> tianshou BasePolicy.compute_episodic_return / _gae_return
                (tianshou 0.5.x, numba; NOT in /root/reference: PARITY UNPINNED).  Restated:
                v_s_ masked by ~terminated; end_flag = terminated|truncated, forced True at every
                sub-buffer's unfinished tail; reverse scan gae = delta + gamma*lambda*(1-end)*gae.
  action_dist   ppo_policy.py:142-179: logvar.clamp(min,max), sigma = exp(logvar)**0.5, Independent(Normal,1)
  ppo_loss      ppo_policy.py:189-242 for one minibatch (norm_adv with UNBIASED std + eps, clip 0.1,
                vf mean squared error without value clip, entropy bonus)
"""
import math

import numpy as np
import torch

_EPS = float(np.finfo(np.float32).eps)  # tianshou BasePolicy._eps


def gae_returns(v_s, v_s_next, rew, terminated, truncated, gamma=0.99, gae_lambda=0.95):
    """Per-env time-major arrays [A, n]: the collector stores env a's transitions contiguously and the
    last stored step of every env is an 'unfinished tail' (end_flag forced True).  numpy float64 like
    tianshou (rew is float64 in its buffers)."""
    v_s = np.asarray(v_s, np.float64)
    v_s_ = np.asarray(v_s_next, np.float64) * (~np.asarray(terminated, bool))
    rew = np.asarray(rew, np.float64)
    end = np.logical_or(terminated, truncated).copy()
    end[:, -1] = True
    A, n = rew.shape
    adv = np.zeros((A, n))
    delta = rew + v_s_ * gamma - v_s
    disc = (1.0 - end) * (gamma * gae_lambda)
    for a in range(A):
        gae = 0.0
        for i in range(n - 1, -1, -1):
            gae = delta[a, i] + disc[a, i] * gae
            adv[a, i] = gae
    return adv + v_s, adv


def action_dist(mu, logvar, min_logvar=-2.5, max_logvar=2.5):
    logvar = logvar.clamp(min_logvar, max_logvar)
    sigma = torch.exp(logvar) ** 0.5
    return mu, sigma


def log_prob(mu, sigma, act):
    """Independent(Normal(mu, sigma), 1).log_prob"""
    var = sigma ** 2
    return (-((act - mu) ** 2) / (2 * var) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))).sum(-1)


def entropy(sigma):
    return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(sigma)).sum(-1)


def ppo_loss(mu, logvar, value, act, adv, returns, logp_old, eps_clip=0.1, vf_coef=1.0, ent_coef=0.01, norm_adv=True,
             dual_clip=None, value_clip=False, v_s=None):
    """ppo_policy.py:189-241; `dual_clip` :204-207, `value_clip` with the old values `v_s` :216-221 (both off in main_ppo.py)."""
    mu_, sigma = action_dist(mu, logvar)
    if norm_adv:
        adv = (adv - adv.mean()) / (adv.std() + _EPS)
    lp = log_prob(mu_, sigma, act)
    ratio = (lp - logp_old).exp().to(lp.dtype)   # reference: .float() - the same in fp32; a float64 evaluation stays float64
    surr1 = ratio * adv
    surr2 = ratio.clamp(1.0 - eps_clip, 1.0 + eps_clip) * adv
    if dual_clip:
        clip1 = torch.min(surr1, surr2)
        clip2 = torch.max(clip1, dual_clip * adv)
        clip_loss = -torch.where(adv < 0, clip2, clip1).mean()
    else:
        clip_loss = -torch.min(surr1, surr2).mean()
    value = value.flatten()
    if value_clip:
        v_clip = v_s + (value - v_s).clamp(-eps_clip, eps_clip)
        vf_loss = torch.max((returns - value).pow(2), (returns - v_clip).pow(2)).mean()
    else:
        vf_loss = (returns - value).pow(2).mean()
    ent = entropy(sigma).mean()
    loss = clip_loss + vf_coef * vf_loss - ent_coef * ent
    return loss, {"loss/clip": clip_loss, "loss/vf": vf_loss, "loss/ent": ent, "loss/kld": 0.5 * torch.mean(mu.pow(2))}


def split_spans(n: int, size: int):
    """tianshou.data.Batch.split(size, merge_last=True) [upstream]: chunks of `size` rows of a permutation, a remainder shorter
    than `size` merged into the last chunk."""
    merge = n % size > 0
    spans, i = [], 0
    while i < n:
        if merge and i + 2 * size >= n:
            spans.append((i, n))
            break
        spans.append((i, min(i + size, n)))
        i += size
    return spans


def ppo_learn(P, optim, clip_params, obs, act, adv, returns, logp_old, batch_size, perms, v_s=None, z_mu=None, repeat=1, eps_clip=0.1,
              vf_coef=1.0, ent_coef=0.01, max_grad_norm=0.1, norm_adv=True, dual_clip=None, value_clip=False, on_step=None):
    """ppo_policy.py:182-265 (`GAMMAPPOPolicy.learn`) on a dict of parameter tensors `P` (keys of the policy's state_dict, leaves
    that require grad; evaluated with oracle.nets.policy_*): per pass one permutation (`perms[pass]`, the rows Batch.split draws),
    per minibatch the loss of :189-241, zero_grad / backward (:242-243), clip_grad_norm_ over `clip_params` - tianshou's
    ActorCritic(actor, critic): `self._actor_critic: ActorCritic(...)` of :88 is an annotation, so the SHARED encoders are not
    clipped (:244-247) - and optim.step (:248); the logged `loss/kld` is 0.5 mean(z_mu^2) of the ROLLOUT's stored means (:232,
    `minibatch.z_mu`); early stop on the last minibatch's mean(logp_old - logp) >= 0.02 (:255-258).
    `on_step(i, grads)` is called before every optimiser step (tests)."""
    from . import nets
    n = act.shape[0]
    out = {"loss": [], "loss/clip": [], "loss/vf": [], "loss/ent": [], "loss/kld": []}
    k = 0
    for step in range(repeat):
        perm = torch.as_tensor(perms[step]).long()
        kl = None
        for s, e in split_spans(n, batch_size):
            i = perm[s:e]
            o = {kk: v[i] for kk, v in obs.items()}
            hx = nets.policy_base(P, o)
            mu, logvar = nets.policy_actor(P, hx)
            value = nets.policy_critic(P, hx)
            loss, terms = ppo_loss(mu, logvar, value, act[i], adv[i], returns[i], logp_old[i], eps_clip=eps_clip, vf_coef=vf_coef,
                                   ent_coef=ent_coef, norm_adv=norm_adv, dual_clip=dual_clip, value_clip=value_clip,
                                   v_s=None if v_s is None else v_s[i])
            optim.zero_grad()
            loss.backward()
            if max_grad_norm:
                torch.nn.utils.clip_grad_norm_(clip_params, max_norm=max_grad_norm)
            if on_step is not None:
                on_step(k, P)
            optim.step()
            k += 1
            out["loss"].append(float(loss.detach()))
            for kk in ("loss/clip", "loss/vf", "loss/ent"):
                out[kk].append(float(terms[kk].detach()))
            out["loss/kld"].append(float(0.5 * torch.mean(z_mu[i].pow(2))) if z_mu is not None else float(terms["loss/kld"].detach()))
            with torch.no_grad():
                m_, s_ = action_dist(mu, logvar)
                kl = float((logp_old[i] - log_prob(m_, s_, act[i])).mean())
        if kl is not None and kl >= 0.02:
            break
    return out
