#!/usr/bin/env python3
"""Development aid: the rollout dense kernel (egx_linear) against the library GEMM (torch.addmm, TunableOp-tuned on the fly)
at the shapes of sample_prior / policy_forward, as dependent chains of 20 launches captured in a HIP graph."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import _lib
lib = _lib.load()
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True); torch.cuda.tunable.set_filename("/tmp/bench_linear_tunable.csv")


def egx(x, w, b, out, act=1):
    d = _lib.LinearDesc()
    d.num_rows, d.out_features, d.num_segments = x.shape[0], w.shape[0], 1
    d.seg_ptr[0], d.seg_width[0], d.seg_ld[0] = x.data_ptr(), x.shape[1], x.shape[1]
    d.weight, d.weight_ld, d.bias = w.data_ptr(), w.shape[1], b.data_ptr()
    d.residual, d.residual_ld = None, 0
    d.out, d.out_ld, d.activation, d.leaky_slope = out.data_ptr(), out.shape[1], act, 0.0
    _lib.check(lib.egx_linear(C.byref(d), _lib.current_stream_ptr()), "egx_linear")


def timed(fn, n=20, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for M in (512, 256):
    for N, K in ((768, 256), (512, 256), (256, 512), (1152, 1152), (1536, 402), (256, 256)):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        # dependent chain: out feeds nothing, but launches are back to back on one stream; square shapes chain for real
        t_egx = timed(lambda: egx(x, w, b, out))
        t_lib = timed(lambda: torch.addmm(b, x, w.t(), out=out))
        print(f"M={M} N={N} K={K}: egx_linear {t_egx:6.2f} us   library {t_lib:6.2f} us   ({2*M*N*K/t_egx/1e6:5.1f} vs {2*M*N*K/t_lib/1e6:5.1f} TFLOP/s)", flush=True)
