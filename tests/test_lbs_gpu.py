"""GPU parity of the SMPL-X / SDF kernels (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from egogen_amd import synth
from tests.helpers import load_golden, max_abs

pytestmark = pytest.mark.gpu


def _band(mode):
    """Distance from the zero level set (metres) inside which a penetration count may differ from the oracle's: 2e-5 = fp32
    round-off of the vertex chain - in EVERY mode.  "f16mix" evaluates the pose-corrective columns of the count-only tiles as one
    fp16 product (~4 um rms, ~22 um worst case), but only to classify: a vertex whose SDF value is closer to zero than that error
    can explain is re-evaluated in fp32 inside the kernel (csrc/body_model.hip: lbs_fix_process), so its counts are held to the
    same band as the fp32-equivalent modes.  Positions (markers, joints, landmarks) are held to 2e-5 m in every mode."""
    return 2e-5


@pytest.fixture(params=["f32", "bf16x3", "bf16x2", "f16mix"])
def blend_mode(request):
    """Every arithmetic mode of the blend GEMM (include/egogen_hip.h: egx_lbs_set_blend_mode)."""
    from egogen_amd import _lib
    lib = _lib.load()
    old = int(lib.egx_lbs_get_blend_mode())
    _lib.check(lib.egx_lbs_set_blend_mode({"f32": 0, "bf16x3": 1, "bf16x2": 2, "f16mix": 3}[request.param]), "egx_lbs_set_blend_mode")
    yield request.param
    _lib.check(lib.egx_lbs_set_blend_mode(old), "egx_lbs_set_blend_mode")


def _setup(V, seed=0):
    from egogen_amd.body_model import BodyModelHandle
    from oracle.smplx_lbs import BodyModel
    bm = synth.make_body_model(seed, num_verts=V)
    mk, feet = synth.marker_ids(V), synth.feet_vids(V)
    return bm, mk, feet, BodyModelHandle(bm, mk, feet), BodyModel(bm)


def _poses(A, T, seed):
    g = torch.Generator().manual_seed(seed)
    B = A * T
    xb = torch.zeros(B, 93)
    xb[:, 0:2] = torch.rand(B, 2, generator=g) * 4 - 2
    xb[:, 2] = 0.8 + 0.3 * torch.rand(B, generator=g)
    xb[:, 3:6] = torch.randn(B, 3, generator=g) * 0.8
    xb[:, 6:69] = torch.randn(B, 63, generator=g) * 0.2
    xb[:, 69:] = torch.randn(B, 24, generator=g) * 0.5
    betas = torch.randn(A, 10, generator=g)
    return xb, betas


@pytest.mark.parametrize("V,A,T", [(2048, 37, 1), (1000, 3, 20), (10475, 2, 20)])
def test_lbs_matches_oracle(V, A, T, blend_mode):
    from oracle.smplx_lbs import smplx_forward
    bm, mk, feet, h, ob = _setup(V)
    xb, betas = _poses(A, T, seed=V + A)
    out = h.forward(xb.cuda(), betas.cuda(), T, want_verts=True)
    torch.cuda.synchronize()
    v, j = smplx_forward(ob, xb, betas.repeat_interleave(T, 0))
    # tolerance: 1e-4 relative (north_star) on metre-scale coordinates -> 2e-5 absolute is far inside it
    assert max_abs(out["vertices"].cpu(), v) < 2e-5
    assert max_abs(out["joints"].cpu(), j) < 2e-5
    assert max_abs(out["markers"].cpu(), v[:, torch.as_tensor(mk).long()]) < 2e-5
    # fused path without the vertex tensor (the hot path; this is where the blend mode applies): same picks
    out2 = h.forward(xb.cuda(), betas.cuda(), T, want_verts=False)
    torch.cuda.synchronize()
    assert max_abs(out2["joints"].cpu(), j) < 2e-5
    assert max_abs(out2["markers"].cpu(), v[:, torch.as_tensor(mk).long()]) < 2e-5
    if blend_mode == "f32":
        assert torch.equal(out2["joints"], out["joints"]) and torch.equal(out2["markers"], out["markers"])
    else:  # the split product agrees with the fp32 MFMA to fp32 round-off
        assert max_abs(out2["markers"].cpu(), out["markers"].cpu()) < 3e-6


def test_lbs_vertices_match_in_tree_skinning_golden(blend_mode):
    """The HIP kernels against tests/golden/lbs_skin_ref.npz = vertices produced by the reference tree's own skinning
    restatement (experiments/HOOD/utils/lbs.py::pose_garment) - not by the oracle."""
    from egogen_amd.body_model import BodyModelHandle
    g = load_golden("lbs_skin_ref.npz")
    V = int(g["num_verts"])
    bm = synth.make_body_model(int(g["body_seed"]), num_verts=V)
    h = BodyModelHandle(bm, synth.marker_ids(V), synth.feet_vids(V))
    xb, betas = torch.from_numpy(g["xb"]).cuda(), torch.from_numpy(g["betas"]).cuda()
    out = h.forward(xb, betas, 1, want_verts=True)
    keep = g["vertex_ids"]               # V = 2048 (64 vertex tiles), 257 translated bodies: the fixture holds 220 vertices of each
    assert max_abs(out["vertices"].cpu()[:, keep], g["verts"]) < 2e-5
    mk = np.asarray(synth.marker_ids(V))
    col = {int(v): i for i, v in enumerate(keep)}
    picks = h.forward(xb, betas, 1)      # the hot path (selected blend mode): markers are rows of the same vertices
    assert max_abs(picks["markers"].cpu(), g["verts"][:, [col[int(v)] for v in mk]]) < 2e-5


@pytest.mark.parametrize("A,T", [(1, 1), (257, 1), (105, 20)])
def test_lbs_ragged_batches(A, T, blend_mode):
    """Single body, one body past a 256-body group, and 2100 bodies (>= 8 body groups: the XCD-partitioned item list with
    a ragged last group) - picks and fused SDF counts against the oracle in both blend modes."""
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import smplx_forward
    V = 1000
    bm, mk, feet, h, ob = _setup(V)
    xb, betas = _poses(A, T, seed=A)
    xb[:, 2] = 0.05
    scene = synth.make_sdf_scene(32)
    out = h.forward(xb.cuda(), betas.cuda(), T, sdf=SdfScene(scene))
    torch.cuda.synchronize()
    v, j = smplx_forward(ob, xb, betas.repeat_interleave(T, 0))
    assert max_abs(out["joints"].cpu(), j) < 2e-5
    assert max_abs(out["markers"].cpu(), v[:, torch.as_tensor(mk).long()]) < 2e-5
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    s = calc_sdf(v, sd)
    s[:, torch.as_tensor(feet).long()] = 1.0  # feet are excluded: neither counted nor "near zero"
    ref, near = s.lt(0).sum(-1), (s.abs() < _band(blend_mode)).sum(-1)
    got = out["pene_count"].cpu().long()
    assert ref.max() > 10
    assert ((got - ref).abs() <= near).all(), (got - ref).abs().max()


def test_lbs_fp64_oracle_agrees():
    """fp64 restatement vs the fp32 HIP path (SURVEY 8(c) invariant)."""
    from oracle.smplx_lbs import BodyModel, smplx_forward
    bm, mk, feet, h, _ = _setup(2048)
    ob64 = BodyModel(bm, dtype=torch.float64)
    xb, betas = _poses(8, 4, seed=5)
    out = h.forward(xb.cuda(), betas.cuda(), 4, want_verts=True)
    v, j = smplx_forward(ob64, xb.double(), betas.double().repeat_interleave(4, 0))
    assert max_abs(out["vertices"].cpu(), v) < 2e-5


def test_lbs_sdf_fused_counts(blend_mode):
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import smplx_forward
    V, A, T = 2048, 5, 20
    bm, mk, feet, h, ob = _setup(V)
    xb, betas = _poses(A, T, seed=11)
    xb[:, 2] = 0.0  # pelvis near the floor: plenty of vertices below z=0 (outside the room)
    scene = synth.make_sdf_scene(48)
    g = torch.Generator().manual_seed(3)
    yaw = torch.rand(A, generator=g) * 6.28
    R0 = torch.zeros(A, 3, 3)
    R0[:, 0, 0], R0[:, 0, 1], R0[:, 1, 0], R0[:, 1, 1], R0[:, 2, 2] = yaw.cos(), -yaw.sin(), yaw.sin(), yaw.cos(), 1.0
    T0 = torch.cat([torch.rand(A, 2, generator=g) * 4 - 2, torch.rand(A, 1, generator=g) * 1.2], -1)
    out = h.forward(xb.cuda(), betas.cuda(), T, want_verts=True, sdf=SdfScene(scene), R0=R0.cuda(), T0=T0.cuda())
    torch.cuda.synchronize()
    v, _ = smplx_forward(ob, xb, betas.repeat_interleave(T, 0))
    vw = torch.einsum("bij,btpj->btpi", R0, v.reshape(A, T, V, 3)) + T0[:, None, None, :]
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    s = calc_sdf(vw.reshape(A * T, V, 3), sd)
    s[:, torch.as_tensor(feet).long()] = 1.0  # feet are excluded: neither counted nor "near zero"
    ref = s.lt(0).sum(-1)
    got = out["pene_count"].cpu().long()
    assert ref.max() > 50, "test should exercise penetration"
    # integer counts: exact except for vertices within fp32 round-off of the zero level set
    near = (s.abs() < _band(blend_mode)).sum(-1)
    assert ((got - ref).abs() <= near).all(), (got - ref).abs().max()
    # and without the vertex write (hot path, selected blend mode)
    out2 = h.forward(xb.cuda(), betas.cuda(), T, sdf=SdfScene(scene), R0=R0.cuda(), T0=T0.cuda())
    got2 = out2["pene_count"].cpu().long()
    assert ((got2 - ref).abs() <= near).all(), (got2 - ref).abs().max()
    if blend_mode == "f32":
        assert torch.equal(out2["pene_count"], out["pene_count"])


def test_lbs_sdf_counts_outside_grid(blend_mode):
    """Bodies partly or completely outside the SDF cube (border clamping; face brackets of the coarse table): counts
    still equal the oracle's calc_sdf on the clamped coordinates."""
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import smplx_forward
    V, A, T = 1000, 12, 4
    bm, mk, feet, h, ob = _setup(V)
    xb, betas = _poses(A, T, seed=21)
    scene = synth.make_sdf_scene(48)  # border samples 1.7 cm beyond the walls: clamped points are strictly "inside"
    R0 = torch.eye(3).repeat(A, 1, 1)
    # agents 0-3 straddle a wall / the ceiling / the grid border, the others are far outside in every direction
    T0 = torch.tensor([[3.6, 0, 0], [0, -3.8, 0.2], [0, 0, 3.9], [3.95, 3.95, 0.0], [9, 0, 0], [-9, 1, 1], [0, 12, 0], [1, -15, 2],
                       [0, 0, 18.0], [0.5, 0.5, -6.0], [7, 7, 7], [-8, -8, -5]], dtype=torch.float32)
    out = h.forward(xb.cuda(), betas.cuda(), T, sdf=SdfScene(scene), R0=R0.cuda(), T0=T0.cuda())
    torch.cuda.synchronize()
    v, _ = smplx_forward(ob, xb, betas.repeat_interleave(T, 0))
    vw = v.reshape(A, T, V, 3) + T0[:, None, None, :]
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    s = calc_sdf(vw.reshape(A * T, V, 3), sd)
    s[:, torch.as_tensor(feet).long()] = 1.0  # feet are excluded: neither counted nor "near zero"
    ref, near = s.lt(0).sum(-1), (s.abs() < _band(blend_mode)).sum(-1)
    got = out["pene_count"].cpu().long()
    assert ref.reshape(A, T)[4:].min() == V - len(feet), "bodies outside the room count as penetrating everywhere"
    assert near.reshape(A, T)[4:].max() == 0
    assert ((got - ref).abs() <= near).all(), (got - ref).abs().max()


def test_calc_sdf_kernel_matches_reference_golden():
    from egogen_amd.utils import calc_sdf
    g = load_golden("calc_sdf_ref.npz")
    for tag in "abc":
        d = {"sdf": torch.from_numpy(g[f"{tag}_sdf"]).cuda(), "center": torch.from_numpy(g[f"{tag}_center"]),
             "scale": torch.from_numpy(g[f"{tag}_scale"])}
        out = calc_sdf(torch.from_numpy(g[f"{tag}_pts"]).cuda(), d).cpu().numpy()
        # grids are N(0,1) with |coords| up to 16: 5e-6 absolute is ~1e-6 relative
        assert max_abs(out, g[f"{tag}_val"]) < 5e-6, tag


def test_calc_sdf_from_bricks_is_bit_identical_to_the_row_major_gather():
    """egx_sdf_sample reads the bricked copy of the grid when the scene's tables exist (every SdfScene) and the row-major grid when
    they do not: the same samples through the same arithmetic - bit-identical values, for points inside, on and far outside the
    cube (border clamp on every face), a grid whose sizes are not multiples of four, NaN / inf coordinates."""
    import ctypes as C
    from egogen_amd import _lib
    from egogen_amd.body_model import SdfScene
    lib = _lib.load()
    g = torch.Generator().manual_seed(12)
    odd = {"sdf": torch.randn(37, 50, 23, generator=g), "center": torch.tensor([0.3, -0.2, 1.0]), "scale": torch.tensor(0.31)}
    for sc in (SdfScene(synth.make_sdf_scene(96)), SdfScene(odd)):
        plain = _lib.SdfGrid()
        C.memmove(C.byref(plain), C.byref(sc.desc), C.sizeof(plain))
        plain.coarse_minmax = None
        pts = (torch.rand(200_003, 3, generator=g) * 12 - 6).cuda()
        pts[:5000] = torch.randn(5000, 3, generator=g).cuda() * 0.4 + torch.tensor([1.5, 0.0, 0.5], device="cuda")
        pts[7] = float("nan"); pts[8, 1] = float("inf"); pts[9, 2] = -float("inf")
        a, b = torch.empty(pts.shape[0], device="cuda"), torch.full((pts.shape[0],), 7.0, device="cuda")
        _lib.check(lib.egx_sdf_sample(C.byref(plain), _lib.ptr(pts), pts.shape[0], _lib.ptr(a), _lib.current_stream_ptr()), "egx_sdf_sample")
        _lib.check(lib.egx_sdf_sample(C.byref(sc.desc), _lib.ptr(pts), pts.shape[0], _lib.ptr(b), _lib.current_stream_ptr()), "egx_sdf_sample")
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and float(a[torch.isfinite(a)].abs().max()) > 0


def test_lbs_invariants_full_size():
    """Size-independent properties at BASELINE config-2 scale (64 agents x 20 frames, V=10475)."""
    bm, mk, feet, h, _ = _setup(10475)
    A, T = 64, 20
    xb, betas = _poses(A, T, seed=1)
    xb, betas = xb.cuda(), betas.cuda()
    o1 = h.forward(xb, betas, T, want_verts=True)
    v1, j1 = o1["vertices"].clone(), o1["joints"].clone()
    # translation equivariance
    d = torch.tensor([0.3, -1.1, 0.7], device="cuda")
    xb2 = xb.clone()
    xb2[:, :3] += d
    o2 = h.forward(xb2, betas, T, want_verts=True)
    assert (o2["vertices"] - d - v1).abs().max() < 1e-5
    assert (o2["joints"] - d - j1).abs().max() < 1e-5
    # zero pose (incl. zero hand mean contribution is NOT zero): pure translation of the rest shape is
    # checked through joint 0: with zero global orient the root joint is the rest root + transl
    xb3 = xb.clone()
    xb3[:, 3:] = 0
    o3 = h.forward(xb3, betas, T, want_verts=True)
    root_rest = o3["joints"][:, 0] - xb3[:, :3]
    per_agent = root_rest.reshape(A, T, 3)
    assert (per_agent - per_agent[:, :1]).abs().max() < 1e-6  # depends on betas only
    # global rotation about the root joint is rigid: pairwise vertex distances are preserved
    xb4 = xb.clone()
    xb4[:, 3:6] = torch.randn(A * T, 3, device="cuda")
    o4 = h.forward(xb4, betas, T, want_verts=True)
    idx = torch.randint(0, 10475, (256,), device="cuda")
    dist1 = (v1[:, idx[:128]] - v1[:, idx[128:]]).norm(dim=-1)
    dist4 = (o4["vertices"][:, idx[:128]] - o4["vertices"][:, idx[128:]]).norm(dim=-1)
    assert (dist1 - dist4).abs().max() < 2e-5


@pytest.mark.parametrize("A", [512, 1100])
def test_lbs_batch_independence_full_size(A, blend_mode):
    """BASELINE scale (512 agents x 20 frames, V = 10475): a body's picks and penetration count do not depend on what else
    is in the launch - the first 37 agents evaluated alone are bit-identical to their rows of the 10240-body launch."""
    from egogen_amd.body_model import SdfScene
    bm, mk, feet, h, _ = _setup(10475)
    T, As = 20, 37   # A = 1100: 22 000 bodies, 86 body groups (ragged last group, uneven split over the 8 XCD chunks)
    xb, betas = _poses(A, T, seed=3)
    xb[:, 2] = 0.3
    xb, betas = xb.cuda(), betas.cuda()
    scene = SdfScene(synth.make_sdf_scene(64))
    g = torch.Generator().manual_seed(9)
    R0 = torch.eye(3).repeat(A, 1, 1).cuda()
    T0 = torch.cat([torch.rand(A, 2, generator=g) * 6 - 3, torch.zeros(A, 1)], -1).cuda()
    full = {k: v.clone() for k, v in h.forward(xb, betas, T, sdf=scene, R0=R0, T0=T0).items()}
    sub = h.forward(xb[:As * T].contiguous(), betas[:As].contiguous(), T, sdf=scene, R0=R0[:As].contiguous(), T0=T0[:As].contiguous())
    torch.cuda.synchronize()
    assert full["pene_count"].max() > 0
    for k in ("joints", "markers", "pene_count"):
        assert torch.equal(full[k][:As * T], sub[k]), k
    assert torch.isfinite(full["joints"]).all() and torch.isfinite(full["markers"]).all()


def test_bad_arguments_raise():
    from egogen_amd import _lib
    bm, mk, feet, h, _ = _setup(1000)
    with pytest.raises(ValueError):
        h.forward(torch.zeros(4, 92, device="cuda"), torch.zeros(4, 10, device="cuda"), 1)
    with pytest.raises(ValueError):
        h.forward(torch.zeros(0, 93, device="cuda"), torch.zeros(0, 10, device="cuda"), 1)
    with pytest.raises(ValueError):
        h.forward(torch.zeros(5, 93, device="cuda"), torch.zeros(2, 10, device="cuda"), 2)


def _dense_weights_model(V, seed, nnz_lo, nnz_hi):
    """Synthetic body whose skinning weights have nnz_lo..nnz_hi non-zeros per vertex (the released SMPL-X weights reach 8-16
    around the torso / hands; synth.make_body_model is exactly 4-nnz) spread over distant joints, so tile joint lists get long."""
    bm = synth.make_body_model(seed, num_verts=V)
    rng = np.random.default_rng(seed + 91)
    W = np.zeros((V, 55), np.float32)
    for v in range(V):
        k = int(rng.integers(nnz_lo, nnz_hi + 1))
        idx = rng.choice(55, size=k, replace=False)
        W[v, idx] = rng.dirichlet(np.ones(k)).astype(np.float32)
    W /= W.sum(1, keepdims=True)
    bm = dict(bm)
    bm["lbs_weights"] = W
    return bm


@pytest.mark.parametrize("nnz", [(8, 16), (20, 55)])
def test_lbs_long_joint_lists(nnz, blend_mode):
    """8-16 (and up to all 55) skinning weights per vertex: the per-tile joint list of the epilogue runs to its full length
    (55 joints), which the 4-nnz synthetic body never exercises."""
    from egogen_amd.body_model import BodyModelHandle, SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import BodyModel, smplx_forward
    V, A, T = 1500, 9, 4
    bm = _dense_weights_model(V, 3, *nnz)
    mk, feet = synth.marker_ids(V), synth.feet_vids(V)
    h, ob = BodyModelHandle(bm, mk, feet), BodyModel(bm)
    assert h.nnz >= nnz[0]
    xb, betas = _poses(A, T, seed=77)
    xb[:, 2] = 0.1
    scene = synth.make_sdf_scene(32)
    out = h.forward(xb.cuda(), betas.cuda(), T, want_verts=True, sdf=SdfScene(scene))
    out2 = h.forward(xb.cuda(), betas.cuda(), T, want_verts=False, sdf=SdfScene(scene))
    torch.cuda.synchronize()
    v, j = smplx_forward(ob, xb, betas.repeat_interleave(T, 0))
    assert max_abs(out["vertices"].cpu(), v) < 2e-5
    assert max_abs(out["joints"].cpu(), j) < 2e-5 and max_abs(out2["joints"].cpu(), j) < 2e-5
    assert max_abs(out2["markers"].cpu(), v[:, torch.as_tensor(mk).long()]) < 2e-5
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    s = calc_sdf(v, sd)
    s[:, torch.as_tensor(feet).long()] = 1.0
    ref, near = s.lt(0).sum(-1), (s.abs() < _band(blend_mode)).sum(-1)
    for o_ in (out, out2):
        assert ((o_["pene_count"].cpu().long() - ref).abs() <= near).all()


def test_lbs_full_size_many_body_groups_vs_oracle(blend_mode):
    """V = 10475 with 2400 bodies = 10 body groups (>= 8: XCD-partitioned, L2-blocked item order with a ragged last group)
    against the ORACLE (not a self-comparison): joints, markers and the fused SDF counts of every body.  The oracle runs in
    chunks to bound host memory."""
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import smplx_forward
    V, A, T = 10475, 120, 20
    bm, mk, feet, h, ob = _setup(V)
    xb, betas = _poses(A, T, seed=2400)
    xb[:, 2] = 0.3
    scene = synth.make_sdf_scene(48)
    out = h.forward(xb.cuda(), betas.cuda(), T, sdf=SdfScene(scene))
    torch.cuda.synchronize()
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    mkl, ftl = torch.as_tensor(mk).long(), torch.as_tensor(feet).long()
    betas_rows = betas.repeat_interleave(T, 0)
    worst_j = worst_m = 0.0
    got = out["pene_count"].cpu().long()
    any_pene = 0
    for s0 in range(0, A * T, 200):
        v, j = smplx_forward(ob, xb[s0:s0 + 200], betas_rows[s0:s0 + 200])
        worst_j = max(worst_j, max_abs(out["joints"][s0:s0 + 200].cpu(), j))
        worst_m = max(worst_m, max_abs(out["markers"][s0:s0 + 200].cpu(), v[:, mkl]))
        s = calc_sdf(v, sd)
        s[:, ftl] = 1.0
        ref, near = s.lt(0).sum(-1), (s.abs() < _band(blend_mode)).sum(-1)
        any_pene = max(any_pene, int(ref.max()))
        assert ((got[s0:s0 + 200] - ref).abs() <= near).all(), (s0, (got[s0:s0 + 200] - ref).abs().max())
    assert worst_j < 2e-5 and worst_m < 2e-5, (worst_j, worst_m)
    assert any_pene > 50


def test_lbs_blend_mode_accuracy_report():
    """Error of every blend mode against the float64 oracle on the same poses (V = 10475): max vertex-pick error and the
    deviation of the fused SDF counts - the numbers quoted in DESIGN.md for the two-plane mode."""
    from egogen_amd import _lib
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import BodyModel, smplx_forward
    lib = _lib.load()
    V, A, T = 10475, 6, 20
    bm, mk, feet, h, _ = _setup(V)
    ob64 = BodyModel(bm, dtype=torch.float64)
    xb, betas = _poses(A, T, seed=99)
    xb[:, 2] = 0.3
    v, j = smplx_forward(ob64, xb.double(), betas.double().repeat_interleave(T, 0))
    scene = synth.make_sdf_scene(48)
    sd = {k: torch.as_tensor(np.asarray(scene[k])).double() for k in ("sdf", "center", "scale")}
    s = calc_sdf(v, sd)
    s[:, torch.as_tensor(feet).long()] = 1.0
    ref, near = s.lt(0).sum(-1), (s.abs() < 2e-5).sum(-1)
    mkl = torch.as_tensor(mk).long()
    rows = []
    old_mode = int(lib.egx_lbs_get_blend_mode())
    try:
        for name, mode in (("f32", 0), ("bf16x3", 1), ("bf16x2", 2), ("f16mix", 3)):
            _lib.check(lib.egx_lbs_set_blend_mode(mode), "mode")
            out = h.forward(xb.cuda(), betas.cuda(), T, sdf=SdfScene(scene))
            torch.cuda.synchronize()
            e_m = max_abs(out["markers"].cpu().double(), v[:, mkl])
            e_j = max_abs(out["joints"].cpu().double(), j)
            dc = (out["pene_count"].cpu().long() - ref).abs()
            band = (s.abs() < _band(name)).sum(-1)
            rows.append((name, e_m, e_j, int(dc.max()), int(dc.sum()), int(band.sum())))
            assert e_m < 2e-5 and e_j < 2e-5 and (dc <= band).all()
    finally:
        _lib.check(lib.egx_lbs_set_blend_mode(old_mode), "mode")
    print("\nmode     max|d marker|  max|d joint|  max|d count|  sum|d count|  vertices within the mode's band of the level set")
    for r in rows:
        print("%-7s  %.2e       %.2e      %5d         %5d         %d" % r)
    # the two-plane mode stays within a factor of a few of fp32 round-off thanks to the template's third term
    assert rows[2][1] < 4 * max(rows[0][1], 1e-6)


@pytest.mark.parametrize("tile", [1, 2])
def test_lbs_mixed_blend_reevaluates_what_its_fp16_product_cannot_decide(tile):
    """Mode 3 ("f16mix") classifies the count-only vertices with one fp16 product and re-evaluates in fp32 those whose SDF value
    lies inside the product's error band (csrc/body_model.hip: lbs_fix_process).  On bodies standing IN the obstacle (thousands of
    counted vertices per body): (i) counts within the 2e-5 m band of the float64 oracle - the band of the fp32 modes; (ii) the
    kernel did re-evaluate vertices, and only a few per thousand of what it counted; (iii) vertices the cheap product alone gets
    wrong exist in this workload (the fp16 product's counts, emulated from the float64 vertices + the product's error, differ
    outside the band), i.e. the test would fail without the fix-up.  Both wave tiles of the kernel: 1 = VALU skinning, 2 = the
    count-only tiles skinned on the matrix pipe in two bf16 planes (lbs_epilogue_cell) - a second, larger classification error
    that the same band absorbs - and the two must give the SAME counts (what either cannot decide goes through the same fp32
    re-evaluation)."""
    from egogen_amd import _lib
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import BodyModel, smplx_forward
    lib = _lib.load()
    V, A, T = 10475, 24, 20
    bm, mk, feet, h, _ = _setup(V)
    ob64 = BodyModel(bm, dtype=torch.float64)
    xb, betas = _poses(A, T, seed=4242)
    xb[:, 2] = 0.3                                       # pelvis 0.3 m above the floor: legs inside it
    scene = synth.make_sdf_scene(64)
    sd = {k: torch.as_tensor(np.asarray(scene[k])).double() for k in ("sdf", "center", "scale")}
    ftl = torch.as_tensor(feet).long()
    old_mode = int(lib.egx_lbs_get_blend_mode())
    try:
        _lib.check(lib.egx_lbs_set_blend_mode(3), "mode")
        _lib.check(lib.egx_lbs_set_wave_tile(tile), "tile")
        sc = SdfScene(scene)
        out = h.forward(xb.cuda(), betas.cuda(), T, sdf=sc)
        torch.cuda.synchronize()
        n_fix = h.fix_stats(A * T)
        _lib.check(lib.egx_lbs_set_wave_tile(3 - tile), "tile")
        other = h.forward(xb.cuda(), betas.cuda(), T, sdf=sc, out={})
        torch.cuda.synchronize()
        assert torch.equal(other["pene_count"], out["pene_count"]), (other["pene_count"] - out["pene_count"]).abs().max()
        assert torch.equal(other["markers"], out["markers"])
        # sub-queues of 4 entries (64 of them): most of the ~1 400 vertices find theirs full and are re-evaluated inside the fused kernel
        # (MFMA-ordered operand images instead of the vertex-major copy: another summation order of the same fp32 values)
        _lib.check(lib.egx_lbs_set_wave_tile(tile), "tile")
        _lib.check(lib.egx_lbs_set_fix_queue_capacity(4), "cap")
        small = h.forward(xb.cuda(), betas.cuda(), T, sdf=sc, out={})
        torch.cuda.synchronize()
        assert n_fix <= h.fix_stats(A * T) <= n_fix + 64 * 64   # (entries reserved in a sub-queue that then proved full count twice)
        assert int((small["pene_count"] != out["pene_count"]).sum()) <= 2
    finally:
        _lib.check(lib.egx_lbs_set_blend_mode(old_mode), "mode")
        _lib.check(lib.egx_lbs_set_wave_tile(0), "tile")
        _lib.check(lib.egx_lbs_set_fix_queue_capacity(0), "cap")
    got = out["pene_count"].cpu().long()
    ref = torch.zeros(A * T, dtype=torch.long)
    near = torch.zeros(A * T, dtype=torch.long)
    for s0 in range(0, A * T, 120):
        v, _j = smplx_forward(ob64, xb[s0:s0 + 120].double(), betas.double().repeat_interleave(T, 0)[s0:s0 + 120])
        s = calc_sdf(v, sd)
        s[:, ftl] = 1.0
        ref[s0:s0 + 120] = s.lt(0).sum(-1)
        near[s0:s0 + 120] = (s.abs() < 2e-5).sum(-1)
    assert int(ref.sum()) > 50_000, int(ref.sum())
    assert ((got - ref).abs() <= near).all(), ((got - ref).abs() - near).max()
    assert ((small["pene_count"].cpu().long() - ref).abs() <= near).all()
    print(f"\nf16mix, wave tile {tile}: {int(ref.sum())} counted vertices, {n_fix} re-evaluated in fp32 ({1e3 * n_fix / max(int(ref.sum()), 1):.2f} per thousand), "
          f"{int(near.sum())} within 2e-5 m of the level set")
    assert 0 < n_fix < 0.02 * int(ref.sum()), n_fix


def test_lbs_tile_lists_skip_only_what_contributes_nothing(blend_mode):
    """Calls without vertex output walk a subset of the 32-vertex tiles: those with a picked vertex (markers / joints only),
    plus those with a vertex of the penetration count (SDF calls; the count excludes the feet).  With the feet chosen as
    whole joint clusters - the kernel sorts vertices by the set of joints they are bound to, so they fill tiles of their own,
    as the feet of the real model do - tiles ARE skipped, and nothing may change: markers / joints equal the all-tiles (vertex-writing) call in
    the same arithmetic, counts equal the recount from its vertices."""
    from egogen_amd.body_model import BodyModelHandle, SdfScene
    from egogen_amd.utils import calc_sdf
    V, A, T = 4096, 3, 20
    bm = synth.make_body_model(0, num_verts=V)
    first = (np.asarray(bm["lbs_weights"]) != 0).argmax(1)                        # lowest joint a vertex is bound to = sort key
    feet = np.flatnonzero(np.isin(first, [7, 8, 9, 10])).astype(np.int32)          # a contiguous run of the kernel's vertex order
    mk = synth.marker_ids(V)
    h = BodyModelHandle(bm, mk, feet)
    assert h.lbs_vertices["picks"] < h.lbs_vertices["sdf"] < V, h.lbs_vertices    # both lists skip something here
    xb, betas = _poses(A, T, seed=5)
    xb[:, 2] = 0.1
    scene = SdfScene(synth.make_sdf_scene(48))
    R0 = torch.eye(3).repeat(A, 1, 1).cuda()
    T0 = torch.zeros(A, 3).cuda()
    full = h.forward(xb.cuda(), betas.cuda(), T, want_verts=True, sdf=scene, R0=R0, T0=T0)
    full = {k: v.clone() for k, v in full.items()}
    picks = h.forward(xb.cuda(), betas.cuda(), T)
    with_sdf = h.forward(xb.cuda(), betas.cuda(), T, sdf=scene, R0=R0, T0=T0)
    if blend_mode == "f32":     # the vertex-writing call always uses the fp32 kernel: same arithmetic -> same bits
        assert torch.equal(picks["markers"], full["markers"]) and torch.equal(picks["joints"], full["joints"])
        assert torch.equal(with_sdf["pene_count"], full["pene_count"])
    else:
        assert float((picks["markers"] - full["markers"]).abs().max()) < 3e-6
        assert float((picks["joints"] - full["joints"]).abs().max()) < 3e-6
    assert torch.equal(with_sdf["markers"], picks["markers"]) and torch.equal(with_sdf["joints"], picks["joints"])
    # recount from the written vertices (feet excluded): exact up to vertices at the level set
    sc = synth.make_sdf_scene(48)
    sd = {k: torch.as_tensor(np.asarray(sc[k])).cuda() for k in ("sdf", "center", "scale")}
    s = calc_sdf(full["vertices"].reshape(A * T, V, 3), sd)
    s[:, torch.as_tensor(feet).long().cuda()] = 1.0
    ref = s.lt(0).sum(-1).cpu()
    near = (s.abs() < _band(blend_mode)).sum(-1).cpu()
    assert ((with_sdf["pene_count"].cpu().long() - ref).abs() <= near).all()
    assert int(with_sdf["pene_count"].max()) > 20


# ---------------------------------------------------------------------------------------------------------------------------
# Free-space culling of SDF work items (csrc/body_model.hip: egx_lbs_cull_kernel): results must be BIT-identical to the launch
# that walks every item.
# ---------------------------------------------------------------------------------------------------------------------------
def _culling(on):
    from egogen_amd import _lib
    _lib.check(_lib.load().egx_lbs_set_culling(1 if on else 0), "egx_lbs_set_culling")


def _random_rotations(n, g, yaw_only=False):
    if yaw_only:
        yaw = torch.rand(n, generator=g) * 6.28
        R = torch.zeros(n, 3, 3)
        R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = yaw.cos(), -yaw.sin(), yaw.sin(), yaw.cos(), 1.0
        return R
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


@pytest.mark.parametrize("V,A,T", [(2048, 12, 20), (2048, 60, 20), (10475, 26, 20)])
@pytest.mark.parametrize("layout", ["in_room", "around_obstacle", "walls_and_outside", "tumbling"])
def test_lbs_culling_is_exact(V, A, T, layout):
    """Culled vs unculled launch on the same inputs: penetration counts, joints and markers equal to the last bit, for bodies
    standing in the room (upper-body tiles provably free: items ARE skipped), bodies placed around / inside the obstacle, bodies
    at the walls, the grid border and far outside (border clamping), and bodies tumbling with arbitrary 3-D frames, extreme
    shapes (betas up to +-4) and large joint rotations (the bound's shape / pose terms)."""
    from egogen_amd.body_model import BodyModelHandle, SdfScene
    bm = synth.make_body_model(0, num_verts=V, structured=True)   # blend shapes with the structure of a learned model: the bound is tight
    mk, feet = synth.marker_ids(V), synth.feet_vids(V)
    h = BodyModelHandle(bm, mk, feet)
    assert h.culls and h.cull_reference_margin < 0.15, h.cull_reference_margin
    g = torch.Generator().manual_seed(V + A)
    xb, betas = _poses(A, T, seed=V + 3 * A)
    scene = synth.make_sdf_scene(128)   # 6 cm voxels, 25 cm bracket cells (the benchmark grid: 3 cm / 12.5 cm)
    sdf = SdfScene(scene)
    R0 = _random_rotations(A, g, yaw_only=True)
    if layout == "in_room":
        xb[:, 0:2] *= 0.1
        xb[:, 3:6] *= 0.2                                      # roughly upright, pelvis ~0.95 m above the floor
        T0 = torch.cat([torch.rand(A, 2, generator=g) * 5 - 2.5, torch.zeros(A, 1)], -1)
        T0[T0[:, 0] > 0.3, 0] -= 3.0                           # away from the obstacle
    elif layout == "around_obstacle":
        xb[:, 0:2] *= 0.1
        T0 = torch.cat([1.5 + (torch.rand(A, 1, generator=g) - 0.5) * 2.4, (torch.rand(A, 1, generator=g) - 0.5) * 2.4,
                        torch.rand(A, 1, generator=g) * 0.4], -1)
    elif layout == "walls_and_outside":
        T0 = (torch.rand(A, 3, generator=g) - 0.5) * torch.tensor([9.0, 9.0, 8.0]) + torch.tensor([0.0, 0.0, 1.0])
        T0[::5] *= 3.0                                          # some far outside the grid
    else:
        R0 = _random_rotations(A, g)
        betas = betas * 4.0 / betas.abs().max()
        xb[:, 6:69] *= 4.0
        xb[:, 69:] *= 3.0
        T0 = (torch.rand(A, 3, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 3.0]) + torch.tensor([0.0, 0.0, 1.5])
    args = (xb.cuda(), betas.cuda(), T)
    kw = dict(sdf=sdf, R0=R0.cuda(), T0=T0.cuda())
    try:
        _culling(False)
        ref = {k: v.clone() for k, v in h.forward(*args, **kw).items()}
        _culling(True)
        out = h.forward(*args, **kw)
        act, tot = h.cull_stats(A * T)
    finally:
        _culling(False)      # the library default
    assert torch.equal(out["pene_count"], ref["pene_count"]), (out["pene_count"].long() - ref["pene_count"].long()).abs().max()
    assert torch.equal(out["joints"], ref["joints"]) and torch.equal(out["markers"], ref["markers"])
    assert 0 < act <= tot
    if layout == "in_room":
        assert act < tot, (act, tot)                           # bodies a metre above the floor, away from the walls: items ARE skipped
    if layout == "around_obstacle":
        assert int(ref["pene_count"].max()) > 100              # the exact path is exercised
    print(f"culling {layout} V={V} A={A}: {act} of {tot} items evaluated, max count {int(ref['pene_count'].max())}")


def test_lbs_culling_thin_wall_between_joints():
    """A one-voxel-thick plate through the middle of standing bodies (between pelvis and chest, between the legs): the hull of
    the balls around two joints spans the plate although neither ball touches it - the box test must keep those items."""
    from egogen_amd.body_model import SdfScene
    from oracle.sdf import calc_sdf
    from oracle.smplx_lbs import smplx_forward
    from egogen_amd.body_model import BodyModelHandle
    from oracle.smplx_lbs import BodyModel
    V, A, T = 2048, 13, 20
    bm = synth.make_body_model(0, num_verts=V, structured=True)
    mk, feet = synth.marker_ids(V), synth.feet_vids(V)
    h, ob = BodyModelHandle(bm, mk, feet), BodyModel(bm)
    assert h.culls
    xb, betas = _poses(A, T, seed=77)
    xb[:, 0:2] *= 0.05
    xb[:, 3:6] *= 0.1
    res = 64
    grid = np.full((res, res, res), -0.5, np.float32)           # free space everywhere ...
    zi = int((1.15 - (1.0 - 4.0)) / 8.0 * res)                  # ... except the sample layer nearest to z = 1.15 m (chest height)
    grid[:, :, zi] = 0.3
    xi = int((0.0 + 4.0) / 8.0 * res)
    grid[xi, :, :] = np.maximum(grid[xi, :, :], 0.2)            # and a vertical plate at x ~ 0 (between the legs of some bodies)
    scene = {"sdf": grid, "center": np.array([0, 0, 1.0], np.float32), "scale": np.float32(0.25)}
    sdf = SdfScene(scene)
    g = torch.Generator().manual_seed(5)
    R0 = _random_rotations(A, g, yaw_only=True)
    T0 = torch.cat([(torch.rand(A, 1, generator=g) - 0.5) * 1.0, (torch.rand(A, 1, generator=g) - 0.5) * 4, torch.zeros(A, 1)], -1)
    args = (xb.cuda(), betas.cuda(), T)
    kw = dict(sdf=sdf, R0=R0.cuda(), T0=T0.cuda())
    try:
        _culling(False)
        ref = {k: v.clone() for k, v in h.forward(*args, **kw).items()}
        _culling(True)
        out = h.forward(*args, **kw)
        act, tot = h.cull_stats(A * T)
    finally:
        _culling(False)      # the library default
    assert torch.equal(out["pene_count"], ref["pene_count"])
    assert int(ref["pene_count"].max()) > 50, "bodies cross the plates"
    # and both equal the oracle's count
    v, _ = smplx_forward(ob, xb, betas.repeat_interleave(T, 0))
    vw = torch.einsum("bij,btpj->btpi", R0, v.reshape(A, T, V, 3)) + T0[:, None, None, :]
    sd = {k: torch.as_tensor(np.asarray(scene[k])) for k in ("sdf", "center", "scale")}
    s = calc_sdf(vw.reshape(A * T, V, 3), sd)
    s[:, torch.as_tensor(feet).long()] = 1.0
    near = (s.abs() < 2e-5).sum(-1)
    assert ((out["pene_count"].cpu().long() - s.lt(0).sum(-1)).abs() <= near).all()
    print(f"thin walls: {act} of {tot} items evaluated")


def test_lbs_culling_is_off_for_the_iid_noise_benchmark_body():
    """The benchmark body of SURVEY 8(d) has i.i.d. noise in every blend-shape entry: the a-priori bound on a vertex's offset
    from its joints is ~0.56 m, nothing is ever provably free, so SDF launches of that model skip the culling kernels
    (`egx_body_model_culls` = 0) - and say so when asked for statistics."""
    from egogen_amd import _lib
    from egogen_amd.body_model import SdfScene
    bm, mk, feet, h, ob = _setup(2048)
    assert not h.culls and h.cull_reference_margin > 0.3, h.cull_reference_margin
    xb, betas = _poses(12, 20, seed=1)
    try:
        _culling(True)     # even when asked for
        h.forward(xb.cuda(), betas.cuda(), 20, sdf=SdfScene(synth.make_sdf_scene(32)), R0=torch.eye(3).repeat(12, 1, 1).cuda(), T0=torch.zeros(12, 3).cuda())
        with pytest.raises(_lib.EgxError):
            h.cull_stats(240)
    finally:
        _culling(False)
