#!/usr/bin/env python3
"""Micro-benchmark of the SMPL-X/SDF kernels alone (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
from egogen_amd import synth, _lib
from egogen_amd.body_model import BodyModelHandle, SdfScene

bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
lib0 = _lib.load()
modes = [int(m) for m in os.environ.get("EGX_BENCH_MODES", "1,2,0").split(",")]
agents = [int(a) for a in os.environ.get("EGX_BENCH_AGENTS", "64,512").split(",")]
for mode, A in [(m, a) for m in modes for a in agents]:
    lib0.egx_lbs_set_blend_mode(mode)
    T = 20
    B = A * T
    g = torch.Generator().manual_seed(0)
    xb = (torch.randn(B, 93, generator=g) * 0.2).cuda(); xb[:, 2] += float(os.environ.get("EGX_BENCH_Z", "1"))   # 0.3: legs inside the floor
    betas = torch.randn(A, 10, generator=g).cuda()
    R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
    for name, kw in (("picks", {}), ("picks+sdf", dict(sdf=scene, R0=R0, T0=T0)),
                     ("verts", dict(want_verts=True)), ("verts+sdf", dict(want_verts=True, sdf=scene, R0=R0, T0=T0))):
        out = {}
        for _ in range(3):
            h.forward(xb, betas, T, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            h.forward(xb, betas, T, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        # the fused kernel alone (HIP events recorded by the library around that launch)
        lib = _lib.load()
        kms = []
        for _ in range(5):
            k0, k1 = C.c_void_p(), C.c_void_p()
            lib.egx_event_create(C.byref(k0)); lib.egx_event_create(C.byref(k1))
            lib.egx_profile_next_lbs(k0, k1)
            h.forward(xb, betas, T, out=out, **kw)
            torch.cuda.synchronize()
            v = C.c_float(); lib.egx_event_elapsed_ms(k0, k1, C.byref(v)); kms.append(v.value)
            lib.egx_event_destroy(k0); lib.egx_event_destroy(k1)
        kms = min(kms)
        nfix = h.fix_stats(B) if (mode == 3 and "sdf" in name and "verts" not in name) else -1
        flops = B * 31425 * 469 * 2
        print(f"mode={mode} A={A} B={B} {name:10s} {ms:8.3f} ms (fused kernel {kms:6.3f} ms = {flops/kms/1e9:6.1f} TF)  blend {flops/ms/1e9:7.1f} TFLOP/s  "
              f"{'verts %.1f GB/s' % (B*10475*12/ms/1e6) if 'verts' in name else ''} fixups {nfix} counted {int(out['pene_count'].sum()) if 'pene_count' in out else -1}", flush=True)
