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
