#!/usr/bin/env python3
"""K2 of SURVEY 8(d): the standalone `calc_sdf` (crowd_ppo/utils.py:54-84 -> egogen_amd.utils.calc_sdf -> egx_sdf_sample_kernel)
at configs[1] scale - 64 agents x 20 frames x 10 475 vertices = 13.4 M points in the 256^3 single-box grid.

Algorithmic bytes: 16 B / point (12 B of coordinates in, 4 B of value out); the 64 MiB grid is a gather target that mostly lives
in L2 / Infinity Cache (a body touches ~1 MiB of it).  Prints one JSON line per point set:
  bodies   the posed vertices of 1280 synthetic bodies standing in the room (what the reference passes)
  uniform  the same number of points drawn uniformly over the cube (no locality: the gather's worst case)
Run under rocprofv3 for the kernel-trace / PMC figures (scripts/run_sdf_profile.sh)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from egogen_amd import synth  # noqa: E402
from egogen_amd.body_model import BodyModelHandle, SdfScene  # noqa: E402
from egogen_amd.utils import calc_sdf  # noqa: E402

A, T = int(os.environ.get("EGX_SDF_AGENTS", 64)), 20
iters = int(os.environ.get("EGX_SDF_ITERS", 50))
bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
g = torch.Generator().manual_seed(0)
B = A * T
xb = (torch.randn(B, 93, generator=g) * 0.2)
xb[:, 0:2] = torch.rand(B, 2, generator=g) * 6 - 3
xb[:, 2] = 1.0
betas = torch.randn(A, 10, generator=g)
verts = h.forward(xb.cuda(), betas.cuda(), T, want_verts=True, want_joints=False, want_markers=False)["vertices"]
uni = torch.rand(B, verts.shape[1], 3, generator=g).cuda() * 8 - 4
uni[..., 2] += 1
import ctypes as C  # noqa: E402
from egogen_amd import _lib  # noqa: E402
lib = _lib.load()


plain = _lib.SdfGrid()          # the same grid without its tables: egx_sdf_sample then gathers from the row-major grid
C.memmove(C.byref(plain), C.byref(scene.desc), C.sizeof(plain))
plain.coarse_minmax = None


def gather(pts, out):
    _lib.check(lib.egx_sdf_sample(C.byref(plain), _lib.ptr(pts), pts.shape[0] * pts.shape[1], _lib.ptr(out), _lib.current_stream_ptr()), "egx_sdf_sample")


def bricks(pts, out):
    _lib.check(lib.egx_sdf_sample(C.byref(scene.desc), _lib.ptr(pts), pts.shape[0] * pts.shape[1], _lib.ptr(out), _lib.current_stream_ptr()), "egx_sdf_sample")


for name, pts in (("bodies", verts), ("uniform", uni)):
    n = pts.shape[0] * pts.shape[1]
    ref = None
    for kname, fn in (("row-major grid (egx_sdf_sample_kernel)", gather), ("4x4x4 bricks (egx_sdf_sample_bricks_kernel)", bricks)):
        out = torch.empty(pts.shape[0], pts.shape[1], device="cuda")
        for _ in range(5):
            fn(pts, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(pts, out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        ref = out.clone() if ref is None else ref
        bad = (out.view(torch.int32) != ref.view(torch.int32))
        if bool(bad.any()):
            idx = bad.nonzero()[:4]
            print("MISMATCH", int(bad.sum()), [(int(r), int(c), float(out[r, c]), float(ref[r, c]), pts[r, c].tolist()) for r, c in idx], file=sys.stderr, flush=True)
        print(json.dumps({"points": name, "kernel": kname, "n": n, "us_per_call": round(us, 1), "algorithmic_MB": round(16 * n / 1e6, 1),
                          "GBps_16B_per_point": round(16 * n / us / 1e3, 1), "frac_of_8TBps": round(16 * n / us / 1e3 / 8000, 3),
                          "negative": int((out < 0).sum()), "bit_identical_to_gathers": bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))}),
              flush=True)
