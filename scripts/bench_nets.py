#!/usr/bin/env python3
"""Micro-benchmark of egx_linear over the shapes of the rollout networks (development aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import _lib
lib = _lib.load()

def run(M, widths, N, act=0, iters=30):
    segs = [torch.randn(M, w, device="cuda") for w in widths]
    K = sum(widths)
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    d = _lib.LinearDesc()
    d.num_rows, d.out_features, d.num_segments = M, N, len(segs)
    for i, s in enumerate(segs):
        d.seg_ptr[i], d.seg_width[i], d.seg_ld[i] = s.data_ptr(), s.shape[1], s.stride(0)
    d.weight, d.weight_ld, d.bias, d.residual, d.residual_ld = W.data_ptr(), 0, b.data_ptr(), None, 0
    d.out, d.out_ld, d.activation, d.leaky_slope = out.data_ptr(), 0, act, 0.01
    st = _lib.current_stream_ptr()
    for _ in range(3):
        lib.egx_linear(C.byref(d), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.egx_linear(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    x = torch.cat(segs, 1)
    ref = torch.nn.functional.linear(x, W, b)
    for _ in range(3):
        torch.nn.functional.linear(x, W, b)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(x, W, b)
    e1.record(); torch.cuda.synchronize()
    us_t = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * M * N * K
    print(f"M={M:6d} K={K:5d} N={N:5d}  egx {us:8.1f} us {fl/us/1e6:7.2f} TF | torch {us_t:8.1f} us {fl/us_t/1e6:7.2f} TF   err {float((out-ref).abs().max()):.1e}", flush=True)

A = 512
print("decode (M=A)")
run(A, [201], 768); run(A, [256], 768); run(A, [256, 128, 201], 768); run(A, [256], 512, 1); run(A, [512], 256, 1); run(A, [256], 201)
print("regressor (M=18A)")
run(18 * A, [201, 159, 10], 128); run(18 * A, [128], 128, 2); run(18 * A, [128], 159)
print("vposer (M=20A)")
run(20 * A, [63], 512, 3); run(20 * A, [512], 512, 3); run(20 * A, [512], 32)
print("policy (M=A and 5A)")
for M in (A, 5 * A):
    run(M, [402], 1536); run(M, [512], 1536); run(M, [32], 1536); run(M, [512, 512, 64, 64], 1152, 3); run(M, [1152], 1152, 3); run(M, [1152], 256); run(M, [1152], 1)
