"""Scene preparation on the GPU: `egx_mesh_sdf` against the CPU check in oracle/mesh_sdf.py and against the
analytic grids the other tests use."""
import numpy as np
import pytest
import torch

from egogen_amd import scene_gen as sg, synth

pytestmark = pytest.mark.gpu


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _octahedron(c, r, R):
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float) * r
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    return v @ R.T + np.asarray(c), f


@pytest.mark.parametrize("inside_positive", [True, False])
def test_mesh_sdf_matches_cpu_check(inside_positive):
    """Two disjoint closed solids in general position (no face parallel to the grid, vertices off the sample planes)."""
    from oracle.mesh_sdf import mesh_signed_distance, sample_positions
    mesh = sg.merge_meshes([_octahedron([0.4, -0.3, 1.1], 1.3, _rot([1, 2, 3], 0.7)),
                            _octahedron([-1.9, 1.6, 0.2], 0.7, _rot([3, -1, 2], 1.9))])
    res, center, half = 28, [0.1, 0.0, 0.9], 3.0
    d = sg.mesh_to_sdf_dict(*mesh, res=res, center=center, half=half, inside_is_obstacle=inside_positive)
    got = d["sdf"].cpu().numpy().astype(np.float64)
    ref = mesh_signed_distance(*mesh, sample_positions(center, 1 / half, res), inside_positive).reshape(res, res, res)
    assert np.abs(np.abs(got) - np.abs(ref)).max() < 2e-6 * half
    far = np.abs(ref) > 1e-5
    assert (np.sign(got[far]) == np.sign(ref[far])).all()
    inside = (ref > 0) == inside_positive
    assert 100 < inside.sum() < inside.size // 2               # both signs are exercised
    assert abs(float(d["scale"]) - 1 / half) < 1e-7 and np.allclose(d["center"].cpu().numpy(), center)


def test_generated_scene_grid_equals_the_analytic_one_and_feeds_calc_sdf():
    """Room shell + box obstacle through the generator == synth.make_sdf_scene (the grid every env test runs on), and the
    result drops into calc_sdf (egx_sdf_sample) unchanged."""
    from egogen_amd.body_model import SdfScene
    from egogen_amd.utils import calc_sdf
    res = 64
    ref = synth.make_sdf_scene(res)
    room = sg.box_mesh(ref["room_lo"], ref["room_hi"])
    obs = sg.box_mesh(ref["obs_lo"], ref["obs_hi"])
    d = sg.scene_sdf_dict(room, obs, res=res, center=ref["center"], half=1.0 / float(ref["scale"]))
    assert float((d["sdf"].cpu() - torch.tensor(ref["sdf"])).abs().max()) < 5e-6
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(2, 500, 3, generator=g) * 8 - 4).cuda()
    pts[..., 2] += 1
    a = calc_sdf(pts, d)
    b = calc_sdf(pts, {k: torch.tensor(ref[k]).cuda() for k in ("sdf", "center", "scale")})
    assert float((a - b).abs().max()) < 5e-6
    assert isinstance(SdfScene(d), SdfScene)


def test_mesh_sdf_full_size_properties():
    """256^3 samples x 2 560 triangles (two subdivided octahedra): |gradient| <= 1 (1-Lipschitz), the zero level set hugs the
    surface, and the count of inside samples matches the solids' volume."""
    def subdiv(v, f, n):
        for _ in range(n):
            nv, nf, mid = list(map(tuple, v)), [], {}
            def m(a, b):
                k = (min(a, b), max(a, b))
                if k not in mid:
                    p = (np.asarray(nv[a]) + np.asarray(nv[b])) / 2
                    mid[k] = len(nv)
                    nv.append(tuple(p))
                return mid[k]
            for a, b, c in f:
                ab, bc, ca = m(a, b), m(b, c), m(c, a)
                nf += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
            v, f = np.asarray(nv), np.asarray(nf)
        return v, f
    v0, f0 = _octahedron([0, 0, 0], 1.0, np.eye(3))
    v0, f0 = subdiv(v0, f0, 4)                                  # 2048 triangles
    v0 = v0 / np.linalg.norm(v0, axis=1, keepdims=True)         # sphere of radius 1 (inscribed polyhedron)
    mesh = sg.merge_meshes([(v0 * 1.5 + [0.3, 0.2, 1.0], f0), _octahedron([-2.6, -2.4, -0.9], 0.8, _rot([1, 1, 0], 0.4)),
                            (v0[:6] * 0 + 100, np.zeros((0, 3), int))])
    mesh = (mesh[0][:len(v0) + 6], mesh[1])
    res, half = 256, 4.0
    d = sg.mesh_to_sdf_dict(*mesh, res=res, center=[0, 0, 1], half=half)
    torch.cuda.synchronize()
    g = d["sdf"]
    h = 2 * half / res
    for ax in range(3):
        assert float((g.diff(dim=ax).abs() / h).max()) <= 1.0 + 1e-3
    vol = float((g > 0).sum()) * h ** 3
    sphere = 4 / 3 * np.pi * 1.5 ** 3
    octa = 4 / 3 * 0.8 ** 3
    assert abs(vol - (sphere + octa)) < 0.02 * (sphere + octa)   # inscribed polyhedron + voxel counting
    assert float(g.max()) < 1.5 + 1e-3 and float(g.min()) < -2.0


def test_prepared_scene_file_drives_the_env_like_the_analytic_scene(tmp_path):
    """N4 -> E1: a scene prepared from meshes (room shell + box obstacle: SDF grid by egx_mesh_sdf, polygon and navmesh from the
    raster, pairs), saved and loaded back through setup_world.build_scene, gives the environment the same rewards, penetration
    counts and egosensing as the analytic single-box scene it models; and the same file as one scene of the box kind drives the
    walkability-map environment."""
    from egogen_amd import setup_world as sw
    from egogen_amd.crowd_env import VecCrowdEnv
    from tests.helpers import build_world
    res = 48
    w = build_world(V=1536, A=6, scene_kind="sdf", sdf_res=res)
    env_a = w["env"]
    ref = w["scene"]
    room = sg.box_mesh(ref["room_lo"], ref["room_hi"])
    obs = sg.box_mesh(ref["obs_lo"], ref["obs_hi"])
    sc = sg.box_scene_from_meshes(ref["room_lo"], ref["room_hi"], obs, radius=0.0, cell=0.1, n_pairs=64, seed=2)
    sdf = sg.scene_sdf_dict(room, obs, res=res, center=ref["center"], half=1.0 / float(ref["scale"]))
    sg.save_scene(str(tmp_path / "my_scene.npz"), sc, sdf)
    loaded = sw.build_scene(str(tmp_path / "my_scene.npz"))
    assert loaded["scene_kind"] == "sdf" and len(loaded["rings"]) == 2
    loaded["pairs"] = w["pairs"]                                        # same start / target pairs as the analytic env
    env_b = VecCrowdEnv(6, w["handle"], env_a.prior, env_a.vposer, seed=0, **loaded)
    assert torch.equal(env_a.pair_valid_mask, env_b.pair_valid_mask)   # the SDF start check agrees pair by pair
    cand = env_a.valid_pairs[:6].reshape(6, 1, 2, 3)
    g = torch.Generator().manual_seed(1)
    for e in (env_a, env_b):
        e.set_candidates(cand)
        e.reset()
    assert float((env_a.obs_ego - env_b.obs_ego).abs().max()) < 2e-4      # polygon from the raster == analytic polygon
    for _ in range(3):
        z = torch.randn(6, 128, generator=g).cuda()
        (oa, ra, ta), (ob, rb, tb) = env_a.step(z, auto_reset=False), env_b.step(z, auto_reset=False)
        assert (env_a.pene_count - env_b.pene_count).abs().max() <= 1     # grids agree to 5e-6: only level-set vertices may flip
        assert float((ra - rb).abs().max()) < 2e-3 and torch.equal(ta, tb)
        assert float((oa["egosensing"] - ob["egosensing"]).abs().max()) < 2e-4
    # the same preparation without an SDF grid = one scene of the box kind (walkability map from the navmesh triangles)
    sg.save_scene(str(tmp_path / "my_box_scene.npz"), sg.box_scene_from_meshes([-4, -4, 0], [4, 4, 0], obs, radius=0.2, cell=0.1,
                                                                              n_pairs=256, seed=3))
    box = sw.build_scene(str(tmp_path / "my_box_scene.npz"))
    assert box["scene_kind"] == "box"
    env_c = VecCrowdEnv(4, w["handle"], env_a.prior, env_a.vposer, seed=1, **box)
    o = env_c.reset()
    o2, r2, t2 = env_c.step(torch.randn(4, 128, generator=g).cuda())
    assert torch.isfinite(r2).all() and torch.isfinite(o2["state"]).all() and set(o["state"].shape) == {4, 2, 402}
