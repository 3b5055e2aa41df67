"""Float64 emulation behind the fix-up threshold of the mixed blend (LBS blend mode 3, csrc/body_model.hip).

The count-only tiles evaluate the pose-corrective columns (k-steps 1..28) as ONE fp16 product.  A vertex whose SDF value is
closer to zero than the error that product can cause is re-evaluated in fp32 by the kernel (fix-up).  This script measures, on
the synthetic body, the actual position error of the fp16 product against

    bound(body) = 2^-11 * sqrt( sum_j ||R_j - I||_F^2 * C_j^2 ),   C_j = max_v max_{k in joint j} |b_k,v|   (3-vector norm)

for ordinary and for wild poses, and prints max / rms of error / bound: the kernel's threshold is KAPPA * bound.
Run: python scripts/emulate_lbs_fixup.py [num_poses]            the fp16 blend product
     python scripts/emulate_lbs_fixup.py skin [num_poses]       the two-plane matrix-pipe skinning (lbs_epilogue_cell): weights and
                                                                cell-mapped joint transforms as hi + mid bf16 planes, products
                                                                hi.hi + hi.mid + mid.hi, against LBS_SKIN_ERR (|v|max + max_j |t_j|)"""
import sys
import numpy as np

sys.path.insert(0, ".")
from egogen_amd import synth  # noqa: E402


def rodrigues(a):
    ang = np.linalg.norm(a + 1e-8, axis=-1, keepdims=True)
    r = a / ang
    K = np.zeros(a.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -r[..., 2], r[..., 1], r[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -r[..., 0], -r[..., 1], r[..., 0]
    s, c = np.sin(ang)[..., None], np.cos(ang)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def bf16_planes(x):
    """hi, mid of egx_bf16_split3 (round to nearest even on the upper 16 bits of the fp32 pattern), as float64"""
    import torch
    t = torch.as_tensor(np.asarray(x, np.float32))
    hi = t.to(torch.bfloat16).to(torch.float32)
    mid = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi.double().numpy(), mid.double().numpy()


def skin(n):
    import torch
    from oracle.smplx_lbs import BodyModel, smplx_forward
    bm = synth.make_body_model(0)
    ob = BodyModel(bm, dtype=torch.float64)
    W = bm["lbs_weights"].astype(np.float64)                       # [V,55]
    vmax = np.linalg.norm(bm["v_template"], axis=1).max() + 0.25
    g = torch.Generator().manual_seed(1)
    Whi, Wmid = bf16_planes(W)
    for name, scale in (("ordinary", 1.0), ("wild (x5)", 5.0)):
        worst_abs = worst_rel = 0.0
        for it in range(n):
            xb = torch.zeros(1, 93, dtype=torch.float64)
            xb[:, 0:2] = torch.rand(1, 2, generator=g, dtype=torch.float64) * 6 - 3
            xb[:, 2] = 1.0
            xb[:, 3:6] = torch.randn(1, 3, generator=g, dtype=torch.float64) * 0.8
            xb[:, 6:69] = torch.randn(1, 63, generator=g, dtype=torch.float64) * 0.2 * scale
            xb[:, 69:] = torch.randn(1, 24, generator=g, dtype=torch.float64) * 0.5 * scale
            betas = torch.randn(1, 10, generator=g, dtype=torch.float64)
            _, _, mid = smplx_forward(ob, xb, betas, return_intermediate=True)
            A = mid["A"][0].numpy()                                # [55,4,4] relative transforms
            vp = mid["v_posed"][0].numpy()                         # [V,3]
            # agent frame: random yaw, single-box scene (8 m cube, 256^3 -> 8 cells per metre), cells = Mc x + const
            yaw = float(torch.rand(1, generator=g)) * 2 * np.pi
            R0 = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
            Mc = 8.0 * R0
            Ap = np.einsum("ar,jrc->jac", Mc, A[:, :3, :])         # [55,3,4]
            exact = np.einsum("vj,jac,vc->va", W, Ap[:, :, :3], vp) + W @ Ap[:, :, 3]
            Ahi, Amid = bf16_planes(Ap)
            T = np.einsum("vj,jac->vac", Whi, Ahi + Amid) + np.einsum("vj,jac->vac", Wmid, Ahi)
            approx = np.einsum("vac,vc->va", T[:, :, :3], vp) + T[:, :, 3]
            err_m = np.linalg.norm(approx - exact, axis=1) / 8.0   # metres
            tmax = np.linalg.norm(A[:, :3, 3], axis=1).max()
            bound = 1.0e-5 * (vmax + tmax)
            worst_abs = max(worst_abs, err_m.max())
            worst_rel = max(worst_rel, err_m.max() / bound)
        print(f"skin {name}: worst position error {worst_abs:.2e} m = {worst_rel:.3f} of LBS_SKIN_ERR (|v|max + max|t|)  ({n} poses x {W.shape[0]} vertices)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "skin":
        return skin(int(sys.argv[2]) if len(sys.argv) > 2 else 24)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    bm = synth.make_body_model(0)
    V = bm["v_template"].shape[0]
    P = bm["posedirs"].astype(np.float64).reshape(54, 9, V, 3)          # joint 1..54
    keep = [j for j in range(1, 55) if not 22 <= j <= 24]               # 51 movable joints
    Pk = P[[j - 1 for j in keep]]                                       # [51, 9, V, 3]
    C = np.sqrt((Pk ** 2).sum(-1)).max(axis=(1, 2))                     # [51] max column norm
    rng = np.random.default_rng(0)
    for name, scale in (("ordinary (0.2 rad body, 0.5 hands)", 1.0), ("wild (x5)", 5.0)):
        worst, rms_acc, cnt = 0.0, 0.0, 0
        worst_abs, fn = 0.0, []
        for _ in range(n):
            aa = np.zeros((55, 3))
            aa[1:22] = rng.normal(0, 0.2 * scale, (21, 3))
            aa[25:55] = rng.normal(0, 0.35 * scale, (30, 3))
            R = rodrigues(aa)
            F = (R[keep] - np.eye(3)).reshape(51, 9)                    # features
            exact = np.einsum("jk,jkvc->vc", F, Pk)
            # fp32 operands rounded to fp16, products exact, fp32-accumulate error ignored (1e-7 relative)
            F16 = F.astype(np.float32).astype(np.float16).astype(np.float64)
            P16 = Pk.astype(np.float32).astype(np.float16).astype(np.float64)
            approx = np.einsum("jk,jkvc->vc", F16, P16)
            err = np.linalg.norm(approx - exact, axis=1)                # [V] metres
            Fj = np.linalg.norm(F, axis=1)
            bound = 2.0 ** -11 * np.sqrt((Fj ** 2 * C ** 2).sum())
            worst = max(worst, err.max() / bound)
            worst_abs = max(worst_abs, err.max())
            rms_acc += (err ** 2).sum() / bound ** 2
            cnt += V
            fn.append(np.linalg.norm(F))
        print(f"{name}: |f|_2 mean {np.mean(fn):.2f}; error max {worst_abs:.2e} m; error/bound max {worst:.3f} rms {np.sqrt(rms_acc / cnt):.3f} "
              f"({cnt} vertex samples)")


if __name__ == "__main__":
    main()
