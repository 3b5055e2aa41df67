"""GPU parity of the network kernels (through the C ABI) vs reference goldens and the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import load_golden, max_abs, rebuild_state_dict

pytestmark = pytest.mark.gpu


def _linear(segs, W, b, act=0, slope=0.0, res=None):
    from egogen_amd import _lib
    lib = _lib.load()
    M, N = segs[0].shape[0], W.shape[0]
    out = torch.empty(M, N, device="cuda")
    d = _lib.LinearDesc()
    d.num_rows, d.out_features, d.num_segments = M, N, len(segs)
    for i, s in enumerate(segs):
        d.seg_ptr[i], d.seg_width[i], d.seg_ld[i] = s.data_ptr(), s.shape[1], s.stride(0)
    d.weight, d.weight_ld, d.bias = W.data_ptr(), 0, (b.data_ptr() if b is not None else None)
    d.residual, d.residual_ld = (res.data_ptr() if res is not None else None), (res.stride(0) if res is not None else 0)
    d.out, d.out_ld, d.activation, d.leaky_slope = out.data_ptr(), 0, act, slope
    _lib.check(lib.egx_linear(C.byref(d), _lib.current_stream_ptr()), "egx_linear")
    return out


@pytest.mark.parametrize("M,widths,N,act", [(5, [201], 768, 0), (64, [256, 128, 201], 768, 1), (333, [201, 159, 10], 128, 2),
                                            (512, [512, 512, 64, 64], 1152, 3), (1, [7], 3, 0), (40, [1152], 1, 0),
                                            # M >= 2048 and N >= 64: the 64x64-tile kernel (ragged M, N, K; segments)
                                            (2048, [63], 512, 3), (2093, [256], 201, 0), (2560, [201, 159, 10], 128, 2),
                                            (4000, [70, 33], 65, 1)])
def test_linear_kernel(M, widths, N, act):
    g = torch.Generator().manual_seed(M + N)
    big = [torch.randn(M, w + 5, generator=g).cuda() for w in widths]   # padded rows: exercise ld != width
    segs = [b[:, 2:2 + w] for b, w in zip(big, widths)]                 # and 8-byte-misaligned starts
    K = sum(widths)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    out = _linear(segs, W, b, act, 0.2, res)
    x = torch.cat([s.double() for s in segs], 1)
    ref = x @ W.double().t() + b.double()
    ref = [ref, torch.tanh(ref), torch.relu(ref), torch.nn.functional.leaky_relu(ref, 0.2)][act] + res.double()
    assert max_abs(out.cpu(), ref.cpu()) < 2e-5


def _combo():
    from egogen_amd.models import GAMMAPrimitiveCombo, PREDICTOR_CFG, REGRESSOR_CFG
    return GAMMAPrimitiveCombo(PREDICTOR_CFG, REGRESSOR_CFG)


def test_state_dict_keys_match_reference():
    from egogen_amd.models import ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, POLICY_CFG
    combo = _combo()
    g = load_golden("cvae_ref.npz")
    assert ["predictor." + str(k) for k in g["state_dict_keys"]] == [k for k in combo.state_dict() if k.startswith("predictor.")]
    g = load_golden("regressor_ref.npz")
    assert ["regressor." + str(k) for k in g["state_dict_keys"]] == [k for k in combo.state_dict() if k.startswith("regressor.")]
    g = load_golden("policy_ref.npz")
    ac = ActorCritic(GAMMAActor(POLICY_CFG), GAMMACritic(POLICY_CFG), GAMMAPolicyBase(POLICY_CFG))
    assert set(str(k) for k in g["state_dict_keys"]) == set(ac.state_dict().keys())
    assert sum(p.numel() for p in ac.parameters()) == 13168001


def test_cvae_decode_matches_reference_golden():
    g = load_golden("cvae_ref.npz")
    sd = rebuild_state_dict(g, [g["fill_seed"]], [""])
    combo = _combo()
    combo.predictor.load_state_dict(sd, strict=True)
    combo.cuda()
    X, z = torch.from_numpy(g["X"]).cuda(), torch.from_numpy(g["z"]).cuda()
    Y, Yb = combo.sample_prior(X, torch.zeros(18, 5, 10, device="cuda"), z)
    assert Y.shape == (18, 5, 201) and Yb.shape == (18, 5, 93)
    assert max_abs(Y.cpu(), g["Y"]) < 1e-4 * max(1.0, np.abs(g["Y"]).max())


def test_sample_prior_matches_oracle():
    from oracle import nets
    g1, g2 = load_golden("cvae_ref.npz"), load_golden("regressor_ref.npz")
    sd = {"predictor." + k: v for k, v in rebuild_state_dict(g1, [g1["fill_seed"]], [""]).items()}
    sd.update({"regressor." + k: v for k, v in rebuild_state_dict(g2, [g2["fill_seed"]], [""], gains=[float(g2["fill_gain"])]).items()})
    combo = _combo()
    combo.load_state_dict(sd, strict=True)
    combo.cuda()
    gen = torch.Generator().manual_seed(7)
    A = 37  # ragged: not a multiple of 32
    X = torch.randn(2, A, 201, generator=gen) * 0.3
    z = torch.randn(A, 128, generator=gen)
    betas = torch.randn(A, 10, generator=gen)
    Y, Yb = combo.sample_prior(X.cuda(), betas[None].repeat(18, 1, 1).cuda(), z.cuda())
    Yo, Ybo = nets.sample_prior(sd, X, betas[None].repeat(18, 1, 1), z)
    assert max_abs(Y.cpu(), Yo) < 1e-4 * max(1.0, float(Yo.abs().max()))
    # the regressed rotations are compared as rotation matrices (axis-angle is discontinuous at pi)
    from oracle.rot import tgm_angle_axis_to_rotation_matrix as aa2R
    assert max_abs(Yb.cpu()[..., :3], Ybo[..., :3]) < 2e-4 * max(1.0, float(Ybo[..., :3].abs().max()))
    assert max_abs(aa2R(Yb.cpu()[..., 3:69].reshape(-1, 3)), aa2R(Ybo[..., 3:69].reshape(-1, 3))) < 2e-4
    assert max_abs(Yb.cpu()[..., 69:], Ybo[..., 69:]) < 2e-4 * max(1.0, float(Ybo[..., 69:].abs().max()))


def test_regressor_row_tile_variants_agree(tmp_path):
    """The fused regressor takes 16, 32 or 48 rows per workgroup depending on the batch (csrc/dense3.hip, egx_launch_regressor3);
    an element's arithmetic does not depend on that, so the three variants (forced with EGX_R3_ROWTILES, one process each: the
    switch is read once) give bit-identical parameters on a ragged batch."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from egogen_amd import setup_world as sw\n"
        "prior = sw.build_motion_prior(seed=0)\n"
        "g = torch.Generator().manual_seed(11)\n"
        "A = 53\n"
        "X = (torch.randn(2, A, 201, generator=g) * 0.3).cuda(); z = torch.randn(A, 128, generator=g).cuda(); b = torch.randn(A, 10, generator=g).cuda()\n"
        "Y, Yb = prior.sample_prior(X, b[None].repeat(18, 1, 1), z)\n"
        "np.save(sys.argv[1], Yb.cpu().numpy())\n" % root)
    outs = []
    for nrt in (1, 2, 3):
        f = str(tmp_path / f"yb{nrt}.npy")
        env = dict(os.environ, EGX_R3_ROWTILES=str(nrt))
        subprocess.run([sys.executable, "-c", code, f], check=True, env=env, cwd=root, timeout=600)
        outs.append(np.load(f))
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[2])


@pytest.mark.parametrize("prec", [0, 2])
def test_policy_matches_reference_golden(prec):
    """egx_policy_forward against the outputs of the reference's own GAMMAPolicyBase / Actor / Critic (policy_ref.npz), in the
    fp32-equivalent arithmetic (three bf16 terms per operand) and in the training drivers' default (two terms, 16 operand
    bits): both inside north_star's 1e-4 relative."""
    from egogen_amd import _lib
    from egogen_amd.models import (ActorCritic, GAMMAActor, GAMMACritic, GAMMAPolicyBase, POLICY_CFG, PolicyHipRunner)
    lib = _lib.load()
    g = load_golden("policy_ref.npz")
    sd = rebuild_state_dict(g, g["fill_seeds"], ["shared_net.", "actor.", "critic."], gains=[1.0, 1.4, 1.4])
    ac = ActorCritic(GAMMAActor(POLICY_CFG), GAMMACritic(POLICY_CFG), GAMMAPolicyBase(POLICY_CFG))
    ac.load_state_dict(sd, strict=True)
    ac.cuda()
    obs = {k[4:]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith("obs_")}
    try:
        _lib.check(lib.egx_policy_set_precision(prec), "egx_policy_set_precision")
        out = PolicyHipRunner(ac.shared_net, ac.actor, ac.critic).forward(obs)
    finally:
        _lib.check(lib.egx_policy_set_precision(0), "egx_policy_set_precision")
    for k, ref in (("mu", g["mu"]), ("logvar", g["logvar"]), ("value", g["value"].reshape(-1))):
        err = max_abs(out[k].cpu(), ref) / max(1.0, np.abs(ref).max())
        print(f"policy forward prec {prec}: {k} max error / scale = {err:.2e}")
        assert err < 1e-4, (k, err)
    # the autograd (update) path computes the same function
    hx = ac.shared_net(obs)
    (mu, lv), _ = ac.actor(hx)
    assert max_abs(mu.detach().cpu(), g["mu"]) < 1e-4 * max(1.0, np.abs(g["mu"]).max())
    assert max_abs(hx.detach().cpu(), g["hx"]) < 2e-5


@pytest.mark.parametrize("n", [1, 47, 100, 1000])
def test_vposer_encoder_matches_oracle(n):
    from egogen_amd.models import VPoserEncoder
    from egogen_amd.synth import seeded_fill
    from oracle import nets
    enc = VPoserEncoder().eval()
    vals = seeded_fill({k: tuple(v.shape) for k, v in enc.state_dict().items()}, 105)
    vals = {k: torch.from_numpy(v) for k, v in vals.items()}
    vals["bodyprior_enc_bn1.num_batches_tracked"] = torch.tensor(0)
    vals["bodyprior_enc_bn2.num_batches_tracked"] = torch.tensor(0)
    enc.load_state_dict(vals)
    enc.cuda()
    x = torch.randn(n, 63, generator=torch.Generator().manual_seed(1)) * 0.3
    out = enc.encode_mean(x.cuda())
    ref = nets.vposer_encode({k: v.float() for k, v in vals.items()}, x)
    assert max_abs(out.cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    # rows inside a wider buffer (the environment hands over the pose columns of its parameter rows)
    wide = torch.full((n, 93), 7.0)
    wide[:, 3:66] = x
    out2 = torch.empty(n, 32, device="cuda")
    wd = wide.cuda()
    enc.encode_mean_into(wd[:, 3:66], 93, n, out2)
    assert torch.equal(out2, out)


def test_vposer_row_tile_variants_agree(tmp_path):
    """egx_vposer3_kernel with 16, 32 or 48 rows per workgroup (EGX_VP_ROWTILES, read once per process): bit-identical output."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from egogen_amd import setup_world as sw\n"
        "vp = sw.build_vposer(seed=0)\n"
        "x = (torch.randn(1001, 63, generator=torch.Generator().manual_seed(3)) * 0.3).cuda()\n"
        "np.save(sys.argv[1], vp.encode_mean(x).cpu().numpy())\n" % root)
    outs = []
    for nrt in (1, 2, 3):
        f = str(tmp_path / f"vp{nrt}.npy")
        subprocess.run([sys.executable, "-c", code, f], check=True, env=dict(os.environ, EGX_VP_ROWTILES=str(nrt)), cwd=root, timeout=600)
        outs.append(np.load(f))
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[2])


def test_policy_bf16_mode_is_close_to_fp32_and_restorable():
    """BASELINE config 5: bf16 operands / fp32 accumulate in the policy's dense layers.  Parity is statistical (SURVEY 8(d)
    C5): over 256 observations the outputs stay within bf16 round-off of the fp32 policy, and switching back restores the
    fp32 results bit for bit."""
    from egogen_amd import _lib
    from egogen_amd.models import GAMMAActor, GAMMACritic, GAMMAPolicyBase, POLICY_CFG, PolicyHipRunner
    lib = _lib.load()
    torch.manual_seed(0)
    shared, actor, critic = GAMMAPolicyBase(POLICY_CFG).cuda(), GAMMAActor(POLICY_CFG).cuda(), GAMMACritic(POLICY_CFG).cuda()
    run = PolicyHipRunner(shared, actor, critic)
    g = torch.Generator().manual_seed(1)
    n = 256
    obs = {"state": (torch.randn(n, 2, 402, generator=g) * 0.3).cuda(), "egosensing": (torch.rand(n, 2, 32, generator=g) * 2 - 1).cuda(),
           "dist": torch.rand(n, generator=g).cuda(), "time": torch.rand(n, generator=g).cuda()}
    ref = {k: v.clone() for k, v in run.forward(obs).items()}
    try:
        _lib.check(lib.egx_policy_set_precision(1), "egx_policy_set_precision")
        assert lib.egx_policy_get_precision() == 1
        low = {k: v.clone() for k, v in run.forward(obs).items()}
    finally:
        _lib.check(lib.egx_policy_set_precision(0), "egx_policy_set_precision")
    again = run.forward(obs)
    for k in ("mu", "logvar", "value"):
        assert torch.equal(again[k], ref[k]), k
        scale = float(ref[k].abs().mean()) + 1e-6
        err = float((low[k] - ref[k]).abs().mean()) / scale
        assert 1e-6 < err < 3e-2, (k, err)   # different from fp32, but by bf16 round-off only


@pytest.mark.parametrize("M,N,K", [(256, 1152, 1152), (37, 201, 585), (512, 159, 128), (1152, 403, 96), (5, 7, 3)])
@pytest.mark.parametrize("trans_a,trans_b", [(False, False), (True, True), (False, True), (True, False)])
def test_gemm3_product_is_fp32_equivalent(M, N, K, trans_a, trans_b):
    """egx_gemm3 (the product behind every training-side autograd node): all operand layouts, ragged shapes, bias /
    activation / residual / saved activation, and in-place accumulation, against float64.  The six-product bf16 split must sit
    at fp32 round-off of the exact result (the bound is that of an fp32 dot product of length K), not at bf16's."""
    from egogen_amd.fused_ops import gemm3
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + 2 * trans_a + trans_b)
    A = torch.randn((K, M) if trans_a else (M, K), generator=g)
    B = torch.randn((K, N) if trans_b else (N, K), generator=g)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    opA, opB = (A.t() if trans_a else A).double(), (B.t() if trans_b else B).double()
    ref = opA @ opB.t()
    bound = 4e-7 * float((opA.abs() @ opB.abs().t()).max()) + 1e-6
    Ad, Bd = A.cuda(), B.cuda()
    assert max_abs(gemm3(Ad, trans_a, Bd, trans_b).cpu(), ref) <= bound
    act_ref = torch.tanh(ref + bias.double())
    saved = torch.empty(M, N, device="cuda")
    out = gemm3(Ad, trans_a, Bd, trans_b, bias=bias.cuda(), act=1, res=res.cuda(), out_act=saved)
    assert max_abs(saved.cpu(), act_ref) <= bound and max_abs(out.cpu(), act_ref + res.double()) <= bound
    acc = res.cuda().clone()
    gemm3(Ad, trans_a, Bd, trans_b, res=acc, out=acc)                  # accumulate in place (weight gradients)
    assert max_abs(acc.cpu(), ref + res.double()) <= bound
    # operands that are row slices of larger tensors (leading dimension > width)
    if not trans_a and not trans_b and M > 8:
        wide = torch.randn(M, K + 5, generator=g).cuda()
        assert max_abs(gemm3(wide[:, :K], False, Bd, False).cpu(), wide[:, :K].cpu().double() @ opB.t()) <= bound
