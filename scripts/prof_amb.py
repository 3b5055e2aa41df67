#!/usr/bin/env python3
"""Development aid: what fraction of body vertices does the SDF bracket table leave undecided in the real loop?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
from egogen_amd import setup_world as sw, synth
from egogen_amd.body_model import BodyModelHandle

A = 512
bm_np, _ = sw.load_body_model()
body = BodyModelHandle(bm_np, synth.marker_ids(), synth.feet_vids())
prior = sw.build_motion_prior(); vposer = sw.build_vposer()
scene = sw.build_scene("single_box", 256)
env = sw.build_env(A, scene, body, prior, vposer, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
for _ in range(3):
    env.step(torch.randn(A, 128, device="cuda", generator=g))
out = {}
body.forward(env.pred_params.reshape(A * 20, 93), env.betas, 20, want_verts=True, out=out)
v = out["vertices"].reshape(A, 20, -1, 3)
world = torch.einsum("aij,atvj->atvi", env.R0.reshape(A, 3, 3), v) + env.T0.reshape(A, 1, 1, 3)
sd = env.sdf
d = sd.desc
c = torch.tensor([d.center[0], d.center[1], d.center[2]], device="cuda")
p = ((world - c) * d.scale + 1) * d.d0 * 0.5 - 0.5
p = p.clamp(0, d.d0 - 1).floor().long() >> 2
tab = sd.coarse.view(torch.float32).reshape(-1, 2)
c1 = (d.d1 + 3) // 4; c2 = (d.d2 + 3) // 4
idx = (p[..., 0] * c1 + p[..., 1]) * c2 + p[..., 2]
mm = tab[idx]
feet = torch.zeros(v.shape[2], dtype=torch.bool, device="cuda"); feet[torch.as_tensor(synth.feet_vids(), device="cuda").long()] = True
und = (~(mm[..., 1] < 0)) & (~(mm[..., 0] > 0)) & (~feet)
ins = (mm[..., 0] > 0) & (~feet)
print("undecided fraction %.4f  decided-inside %.5f  z range of world verts %.2f..%.2f" % (und.float().mean().item(), ins.float().mean().item(), world[..., 2].min().item(), world[..., 2].max().item()))
per_tile = und.float().mean((0, 1)).reshape(-1)[: (v.shape[2] // 32) * 32].reshape(-1, 32).mean(1)
print("vertex tiles with >1%% undecided: %d of %d; mean over those %.3f" % ((per_tile > 0.01).sum().item(), per_tile.numel(), per_tile[per_tile > 0.01].mean().item()))
