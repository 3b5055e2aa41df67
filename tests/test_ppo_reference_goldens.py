"""GAMMAPPOPolicy.learn against the REFERENCE'S OWN `learn`, executed: tests/golden/ppo_learn_ref.npz holds what
crowd_ppo/ppo_policy.py:182-265 did (scripts/gen_env_goldens.py learn: the reference class over a stand-in for the tianshou
parent's constructor and Batch) on 160 transitions, minibatches of 64 and 96 rows (merge_last), one pass: the five loss lists,
and for every optimiser step the gradient each parameter had when `optim.step` ran - AFTER the reference's clip, which covers
actor + critic only (`self._actor_critic: ...` of :88 is an annotation; the shared encoders keep their raw gradients) - and the
parameters afterwards.  Cases: main_ppo.py's defaults; value_clip + dual_clip.

  CPU: oracle/ppo.py::ppo_learn (+ oracle/nets.py) reproduces all of it in float32.
  GPU: egogen_amd.ppo_policy.GAMMAPPOPolicy.learn - hand-written chain replayed as graphs (default case), autograd nodes
       (options case) - reproduces the losses and the parameters after the two AdamW steps.
"""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden, rebuild_state_dict

CASES = ["default", "options"]


def _inputs(g, pre):
    sd = rebuild_state_dict({"state_dict_keys": g[pre + "state_dict_keys"], "state_dict_shapes": g[pre + "state_dict_shapes"]},
                            g[pre + "fill_seeds"], ["shared_net.", "actor.", "critic."], gains=[float(v) for v in g[pre + "fill_gains"]])
    for k in sd:                                     # the generator's last-policy-layer scaling (main_ppo.py:128-131 style)
        if k.startswith("actor.pnet.") and k.endswith(".weight"):
            sd[k] = sd[k] * float(g[pre + "pnet_scale"])
    obs = {k: torch.from_numpy(g[pre + "obs_" + k]) for k in ("state", "egosensing", "dist", "time")}
    t = lambda k: torch.from_numpy(g[pre + k])
    return sd, obs, t("act"), t("adv"), t("returns"), t("logp_old"), t("v_s"), t("z_mu")


def _names(g, pre):
    return [str(n) for n in g[pre + "param_names"]]


@pytest.mark.parametrize("case", CASES)
def test_oracle_ppo_learn_matches_reference_execution(case):
    from oracle import ppo as oppo
    g = load_golden("ppo_learn_ref.npz")
    pre = case + "_"
    sd, obs, act, adv, ret, lpo, v_s, z_mu = _inputs(g, pre)
    names = _names(g, pre)            # ActorCritic.parameters(): actor, critic, shared_net
    assert names[0].startswith("actor.") and names[-1].startswith("shared_net.")
    P = {k: sd[k].clone().requires_grad_(True) for k in names}
    p0 = {k: v.detach().clone() for k, v in P.items()}
    opt = torch.optim.AdamW([P[k] for k in names], lr=3e-4, weight_decay=0.01)
    clip = [P[k] for k in names if not k.startswith("shared_net.")]
    seen = []
    res = oppo.ppo_learn(P, opt, clip, obs, act, adv, ret, lpo, int(g[pre + "batch_size"]), [g[pre + "perm"]], v_s=v_s, z_mu=z_mu,
                         dual_clip=float(g[pre + "dual_clip"]) or None, value_clip=bool(g[pre + "value_clip"]),
                         on_step=lambda i, P_: seen.append({k: v.grad.detach().clone() for k, v in P_.items()}))
    assert [e - s for s, e in oppo.split_spans(160, 64)] == [64, 96]
    for k, v in res.items():
        ref = g[pre + "res_" + k.replace("/", "_")]
        np.testing.assert_allclose(v, ref, rtol=1e-4, atol=2e-6, err_msg=k)
    assert len(seen) == 2
    for i, gr in enumerate(seen):
        nrm = np.array([float(gr[k].norm()) for k in names])
        np.testing.assert_allclose(nrm, g[f"{pre}mb{i}_grad_norm"], rtol=2e-3, atol=1e-7)
        for j, k in enumerate(names):
            scale = float(gr[k].abs().max()) + 1e-12
            assert np.abs(np.resize(gr[k].flatten()[:8].numpy(), 8) - g[f"{pre}mb{i}_grad_head"][j]).max() <= 2e-3 * scale, (i, k)
        clipped = float(torch.sqrt(sum(gr[k].pow(2).sum() for k in names if not k.startswith("shared_net."))))
        assert clipped == pytest.approx(float(g[f"{pre}mb{i}_clipped_set_norm"]), rel=1e-4) and clipped == pytest.approx(0.1, rel=1e-4)
        raw_shared = float(torch.sqrt(sum(gr[k].pow(2).sum() for k in names if k.startswith("shared_net."))))
        assert raw_shared > 1.0          # the encoders' gradient reaches AdamW unclipped (the quirk of :88 / :244-247)
    # parameters after the two steps: AdamW moves an entry by ~lr per step whatever its gradient's size, so an entry whose
    # gradient is within round-off of zero may go the other way - bounded by 2 steps x 2 lr, and rare
    for j, k in enumerate(names):
        d = (P[k].detach() - p0[k])
        assert float(d.norm()) == pytest.approx(float(g[pre + "delta_norm"][j]), rel=2e-2, abs=1e-7), k
        dh = np.abs(np.resize(d.flatten()[:8].numpy(), 8) - g[pre + "delta_head"][j])
        assert dh.max() <= 4 * 3e-4 + 1e-7 and np.median(dh) <= 2e-6, (k, dh)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16x2"])
@pytest.mark.parametrize("case", CASES)
def test_hip_ppo_learn_matches_reference_execution(case, mode):
    from egogen_amd import setup_world as sw
    from egogen_amd.ppo_policy import RolloutBatch
    g = load_golden("ppo_learn_ref.npz")
    pre = case + "_"
    sd, obs, act, adv, ret, lpo, v_s, z_mu = _inputs(g, pre)
    names = _names(g, pre)

    class A:
        seed = 0; lr = 3e-4; gamma = 0.99; gae_lambda = 0.95; max_grad_norm = 0.1; vf_coef = 1.0; ent_coef = 0.01
        weight_kld = 0; rew_norm = False; eps_clip = 0.1; norm_adv = 1; recompute_adv = 0; deterministic_eval = False
        value_clip = int(g[pre + "value_clip"]); dual_clip = float(g[pre + "dual_clip"]) or None
        update_graph = True; update_precision = mode
    pol = sw.build_policy(A())
    full = dict(pol.state_dict())
    for k, v in sd.items():
        full[k] = v
        if k.startswith(("actor.", "critic.")):
            full["_actor_critic." + k] = v
    pol.load_state_dict(full)
    p0 = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    N = 160
    b = RolloutBatch(1, N, "cuda")
    b.state[0].copy_(obs["state"]); b.ego[0].copy_(obs["egosensing"]); b.dist[0].copy_(obs["dist"]); b.time[0].copy_(obs["time"])
    b.act[0].copy_(act); b.adv[0].copy_(adv); b.returns[0].copy_(ret); b.logp_old[0].copy_(lpo); b.mu[0].copy_(z_mu)
    b.values[0].copy_(v_s)
    pol._perm_queue = [g[pre + "perm"]]
    # what each optimiser step consumed: the flat gradient as the backward left it and the clip coefficient egx_adamw_clip_step
    # computed from its [0, n_clip) prefix and applied to that prefix only (csrc/ppo.hip: consts[0], 1024 floats into its
    # workspace) - read after every minibatch, graph replay or not
    seen = []

    def spy(i):
        torch.cuda.synchronize()
        seen.append((pol._flat_grad.detach().cpu().clone(), float(pol._adamw_ws[1024].item()) if pol._flat_opt_state == "ready" else None))
    pol._after_minibatch = spy
    res = pol.learn(b, int(g[pre + "batch_size"]), 1)
    pol._after_minibatch = None
    expect = {"default": {"chain+graph": 1, "chain": 1}, "options": None}[case]
    if expect is not None:   # 64 rows: replayed graph; the merged 96-row minibatch: the chain, eagerly (its own handle)
        assert pol.update_paths == expect, pol.update_paths
    else:
        assert all(k.startswith("autograd") for k in pol.update_paths), pol.update_paths
    tol = 1e-4   # north_star's tolerance, in both arithmetic modes of the update
    for k, v in res.items():
        ref = g[pre + "res_" + k.replace("/", "_")]
        np.testing.assert_allclose(v, ref, rtol=tol, atol=1e-5, err_msg=f"{k} ({mode})")
    sd1 = pol.state_dict()
    for j, k in enumerate(names):
        d = (sd1[k] - p0[k]).cpu()
        assert float(d.norm()) == pytest.approx(float(g[pre + "delta_norm"][j]), rel=3e-2, abs=1e-7), k
        dh = np.abs(np.resize(d.flatten()[:8].numpy(), 8) - g[pre + "delta_head"][j])
        assert dh.max() <= 4 * 3e-4 + 1e-7 and np.median(dh) <= 5e-6, (k, dh)
    # the reference's own gradients at `optim.step` (ppo_policy.py:242-248: AFTER its clip, which covers actor + critic only)
    lay = {id(p_): (off, n) for p_, off, n in pol._flat_layout()}
    named = pol.state_dict(keep_vars=True)
    n_clip = pol._n_clip if pol._flat_opt_state == "ready" else None
    assert len(seen) == 2
    gtol = 2e-3                                # per-tensor gradient norms and leading entries (the CPU oracle's tolerance)
    for i, (flat, coef) in enumerate(seen):
        if n_clip is None:                     # autograd nodes + torch's clip (options case): .grad was scaled in place
            coef, used = 1.0, flat
        else:
            assert 0.0 < coef < 1.0, coef      # the raw actor + critic norm is far above max_grad_norm = 0.1
            used = flat.clone()
            used[:n_clip] *= coef
            # the clipped prefix ends exactly where the first shared_net tensor starts: the encoders are NOT rescaled
            first_shared = min(lay[id(named[k])][0] for k in names if k.startswith("shared_net."))
            last_clipped = max(lay[id(named[k])][0] + lay[id(named[k])][1] for k in names if not k.startswith("shared_net."))
            assert last_clipped <= n_clip <= first_shared, (last_clipped, n_clip, first_shared)
        sl = lambda k: used[lay[id(named[k])][0]:lay[id(named[k])][0] + lay[id(named[k])][1]]
        nrm = np.array([float(sl(k).norm()) for k in names])
        refn = g[f"{pre}mb{i}_grad_norm"]
        print(f"{case}/{mode} step {i}: clip coefficient {coef:.6f}, worst per-parameter |norm / reference - 1| = "
              f"{float(np.max(np.abs(nrm - refn) / np.maximum(refn, 1e-12))):.2e}")
        np.testing.assert_allclose(nrm, refn, rtol=gtol, atol=1e-7, err_msg=f"step {i} per-parameter gradient norms")
        # leading entries of every tensor, relative to the tensor's largest entry: fp32 chain 2e-3 (the CPU oracle's tolerance);
        # the two-term bf16 split (2^-16 per operand) is held to 1e-2 on single entries - its per-tensor norms above are at 2e-4
        htol = gtol if mode == "f32" else 1e-2
        worst_h = 0.0
        for j, k in enumerate(names):
            scale = float(sl(k).abs().max()) + 1e-12
            worst_h = max(worst_h, float(np.abs(np.resize(sl(k)[:8].numpy(), 8) - g[f"{pre}mb{i}_grad_head"][j]).max()) / scale)
        print(f"{case}/{mode} step {i}: worst leading-entry error / largest entry of its tensor = {worst_h:.2e} (bound {htol:g})")
        assert worst_h <= htol
        clipped = float(torch.sqrt(sum(sl(k).double().pow(2).sum() for k in names if not k.startswith("shared_net."))))
        assert clipped == pytest.approx(float(g[f"{pre}mb{i}_clipped_set_norm"]), rel=1e-3) and clipped == pytest.approx(0.1, rel=1e-3)
        raw_shared = float(torch.sqrt(sum(sl(k).double().pow(2).sum() for k in names if k.startswith("shared_net."))))
        ref_shared = float(np.sqrt(sum(float(g[f"{pre}mb{i}_grad_norm"][j]) ** 2 for j, k in enumerate(names) if k.startswith("shared_net."))))
        assert raw_shared > 1.0 and raw_shared == pytest.approx(ref_shared, rel=gtol)
