#!/usr/bin/env python3
"""Development probe for the two-shard collector: the fused LBS launch of one shard on a CU-masked stream beside the policy +
motion-prior chain of the other shard on an ordinary stream, against the same launches alone / back to back.
    python scripts/overlap_probe2.py [agents_per_shard]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from egogen_amd import _lib, setup_world as sw, synth
from egogen_amd.body_model import BodyModelHandle, SdfScene

A = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 20
lib = _lib.load()
bm = synth.make_body_model(0)
h = BodyModelHandle(bm, synth.marker_ids(), synth.feet_vids())
scene = SdfScene(synth.make_sdf_scene(256))
prior = sw.build_motion_prior(seed=0)
vposer = sw.build_vposer(seed=0)
g = torch.Generator().manual_seed(0)
xb = (torch.randn(A * T, 93, generator=g) * 0.2).cuda(); xb[:, 2] += 1
betas = torch.randn(A, 10, generator=g).cuda()
R0 = torch.eye(3).repeat(A, 1, 1).cuda(); T0 = (torch.rand(A, 3, generator=g) * 2 - 1).cuda(); T0[:, 2] = 0
X = (torch.randn(A, 2, 402, generator=g) * 0.3).cuda()
z = torch.randn(A, 128, generator=g).cuda()
Y = torch.empty(18, A, 201, device="cuda"); Yb = torch.empty(18, A, 93, device="cuda")


def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    _lib.check(lib.egx_stream_create_cu_mask(words, 8, C.byref(out)), "egx_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


out_sdf = {}
chain_stream = torch.cuda.Stream()


def lbs_on(stream):
    def f():
        with torch.cuda.stream(stream):
            h.forward(xb, betas, T, out=out_sdf, sdf=scene, R0=R0, T0=T0)
    return f


def chain():
    with torch.cuda.stream(chain_stream):
        prior.sample_prior_into(X[:, 0], X[:, 1], 804, betas, z, Y, Yb)


def both(stream):
    l = lbs_on(stream)
    def f():
        l(); chain()
    return f


plain = torch.cuda.Stream()
print(f"agents per shard {A}: chain alone {timeit(chain):.3f} ms; LBS alone on a plain stream {timeit(lbs_on(plain)):.3f} ms; "
      f"both, two plain streams {timeit(both(plain)):.3f} ms", flush=True)
for name, pred in (("bits 0..223", lambda i: i < 224), ("bits 0..191", lambda i: i < 192), ("i%8 != 7", lambda i: i % 8 != 7),
                   ("i%8 < 6", lambda i: i % 8 < 6), ("i%4 != 3", lambda i: i % 4 != 3)):
    s = masked_stream(pred)
    print(f"  LBS mask {name:12s}: LBS alone {timeit(lbs_on(s)):.3f} ms; LBS + chain concurrently {timeit(both(s)):.3f} ms", flush=True)
