#!/usr/bin/env python3
"""Development aid: a bound on what a split-K form of the 256 x 1152 x 1152 update layer can reach.  The
dense kernel of the chain (through egx_gemm3: 32 x 32 tiles, three-plane arithmetic) on the full problem and on K / 2, K / 4 -
the K / s launch is what ONE of s split-K partial launches does (same tile count, a 1/s of the operand stream), BEFORE the
in-launch reduction seam.  Run under rocprofv3 --kernel-trace --stats, one process per K:
    rocprofv3 --kernel-trace --stats -d /tmp/pk -o p -- python scripts/prof_splitk_bound.py 288"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd.fused_ops import gemm3
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1152
M, N = 256, 1152
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) * 0.05).cuda(); b = torch.randn(N, generator=g).cuda()
out = torch.empty(M, N, device="cuda")
for _ in range(60):
    gemm3(x, False, w, False, bias=b, act=3, slope=0.01, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    gemm3(x, False, w, False, bias=b, act=3, slope=0.01, out=out)
e1.record(); torch.cuda.synchronize()
print(f"K={K}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per gemm3 call (pack launch + dense launch)")
