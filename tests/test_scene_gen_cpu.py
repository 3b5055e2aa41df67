"""Scene preparation, host side: walkable raster -> navmesh triangles, polygon rings, start/target pairs."""
import numpy as np

from egogen_amd import scene_gen as sg, synth
from egogen_amd.egobody import read_ply


def _area(r):
    return abs(0.5 * np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))


def test_box_scene_polygon_navmesh_and_pairs():
    obs = sg.box_mesh([1.0, -0.5, 0.0], [2.0, 0.5, 1.0])
    sc = sg.box_scene_from_meshes([-4, -4, 0], [4, 4, 0], obs, radius=0.0, cell=0.1, n_pairs=500, min_dist=1.7, seed=1)
    rings = sc["rings"]
    assert len(rings) == 2
    # exterior = the floor, hole = the box footprint (cell-aligned here), collinear staircase points removed
    assert sorted(map(tuple, np.round(rings[0][:-1], 6))) == [(-4.0, -4.0), (-4.0, 4.0), (4.0, -4.0), (4.0, 4.0)]
    assert sorted(map(tuple, np.round(rings[1][:-1], 6))) == [(1.0, -0.5), (1.0, 0.5), (2.0, -0.5), (2.0, 0.5)]
    tri = sc["tris"].reshape(-1, 3, 2)
    area = 0.5 * np.abs((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1]) - (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1]))
    assert abs(area.sum() - (64.0 - 1.0)) < 1e-4 and len(tri) <= 16          # merged rectangles, not one pair per cell
    # the triangle cover and the polygon describe the same set (the two walkability tests of the reference: get_map / shapely)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-4.5, 4.5, (4000, 2))
    in_poly = sg.rings_contain(rings, pts[:, 0], pts[:, 1])
    d1 = (pts[:, None, 0] - tri[None, :, 1, 0]) * (tri[None, :, 0, 1] - tri[None, :, 1, 1]) - (tri[None, :, 0, 0] - tri[None, :, 1, 0]) * (pts[:, None, 1] - tri[None, :, 1, 1])
    d2 = (pts[:, None, 0] - tri[None, :, 2, 0]) * (tri[None, :, 1, 1] - tri[None, :, 2, 1]) - (tri[None, :, 1, 0] - tri[None, :, 2, 0]) * (pts[:, None, 1] - tri[None, :, 2, 1])
    d3 = (pts[:, None, 0] - tri[None, :, 0, 0]) * (tri[None, :, 2, 1] - tri[None, :, 0, 1]) - (tri[None, :, 2, 0] - tri[None, :, 0, 0]) * (pts[:, None, 1] - tri[None, :, 0, 1])
    neg = (d1 < 0) | (d2 < 0) | (d3 < 0)
    pos = (d1 > 0) | (d2 > 0) | (d3 > 0)
    in_nav = (~(neg & pos)).any(1)                                             # batch_gen_amass.py:949-961
    assert (in_poly == in_nav).mean() > 0.999
    p = sc["pairs"]
    assert p.shape == (500, 2, 3) and (np.linalg.norm(p[:, 0] - p[:, 1], axis=-1) >= 1.7).all()
    assert sg.rings_contain(rings, p[:, 0, 0], p[:, 0, 1]).all() and sg.rings_contain(rings, p[:, 1, 0], p[:, 1, 1]).all()
    # the analytic box scene of the synthetic generator is the same polygon
    ref = synth.rings_to_edges([synth.rect_ring([-4, -4], [4, 4], True), synth.rect_ring([1.0, -0.5], [2.0, 0.5], False)])
    key = lambda e: {tuple(sorted([tuple(np.round(r[:2], 6)), tuple(np.round(r[2:], 6))])) for r in e}
    assert key(sc["edges"]) == key(ref)


def test_inflation_band_and_largest_component():
    # a wall that cuts the floor in two unequal parts + body radius: the smaller part is dropped, the hole grows by the radius
    wall = sg.box_mesh([1.0, -4.0, 0.0], [1.2, 4.0, 2.5])
    free, origin, cell = sg.walkable_grid([-4, -4, 0], [4, 4, 0], *wall, radius=0.2, cell=0.1)
    rings = sg.grid_to_rings(free, origin, cell)
    assert len(rings) == 1
    xs = rings[0][:, 0]
    assert abs(xs.min() + 4.0) < 1e-9 and abs(xs.max() - 0.8) < 0.1 + 1e-9     # left part, stops `radius` before the wall
    both = sg.grid_to_rings(free, origin, cell, largest_only=False)
    assert len(both) == 2 and _area(both[0]) > _area(both[1])
    # triangles above the height band do not block
    lamp = sg.box_mesh([-1, -1, 2.2], [1, 1, 2.4])
    free2, _, _ = sg.walkable_grid([-4, -4, 0], [4, 4, 0], *lamp, radius=0.2, cell=0.1)
    assert free2.all()


def test_ply_roundtrip(tmp_path):
    obs = sg.box_mesh([0, 0, 0], [1, 2, 3])
    sg.write_ply(str(tmp_path / "n.ply"), *obs)
    v, f = read_ply(str(tmp_path / "n.ply"))
    assert np.allclose(v, obs[0]) and np.array_equal(f, obs[1])
    # outward orientation: positive signed volume
    t = obs[0][obs[1]]
    assert abs(np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6 - 6.0) < 1e-9


def test_oracle_mesh_sdf_against_closed_forms():
    from oracle.mesh_sdf import mesh_signed_distance, sample_positions
    lo, hi = np.array([1.0, -0.5, 0.0]), np.array([2.0, 0.5, 1.0])
    P = sample_positions([0, 0, 1], 1 / 4.0, 12).reshape(-1, 3)
    got = mesh_signed_distance(*sg.box_mesh(lo, hi), P, inside_positive=True)
    ref = -synth._box_sdf(P, lo, hi)            # _box_sdf < 0 inside
    assert np.abs(got - ref).max() < 1e-9


def test_outline_and_triangle_cover_agree_with_the_raster_on_random_grids():
    """Property: for any free / blocked raster the rings (even-odd) and the rectangle cover contain exactly the free cells -
    including diagonal touches, holes inside holes and one-cell corridors."""
    rng = np.random.default_rng(5)
    for trial in range(12):
        nx, ny = int(rng.integers(3, 14)), int(rng.integers(3, 14))
        free = rng.random((nx, ny)) < (0.45 + 0.4 * rng.random())
        if not free.any():
            continue
        origin, cell = np.array([-1.5, 2.0]), 0.25
        rings = sg.grid_to_rings(free, origin, cell, largest_only=False)
        cx = origin[0] + (np.arange(nx) + 0.5) * cell
        cy = origin[1] + (np.arange(ny) + 0.5) * cell
        X, Y = np.meshgrid(cx, cy, indexing="ij")
        inside = sg.rings_contain(rings, X.ravel(), Y.ravel()).reshape(nx, ny)
        assert np.array_equal(inside, free), trial
        assert all(np.array_equal(r[0], r[-1]) and len(r) >= 5 for r in rings)
        v, f = sg.grid_to_navmesh(free, origin, cell)
        tri = v[f][:, :, :2]
        area = 0.5 * np.abs((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1]) - (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1]))
        assert abs(area.sum() - free.sum() * cell * cell) < 1e-9
        # every cell centre is covered by a triangle iff the cell is free
        p = np.stack([X.ravel(), Y.ravel()], 1)
        def side(a, b):
            return (p[:, None, 0] - b[None, :, 0]) * (a[None, :, 1] - b[None, :, 1]) - (a[None, :, 0] - b[None, :, 0]) * (p[:, None, 1] - b[None, :, 1])
        d1, d2, d3 = side(tri[:, 0], tri[:, 1]), side(tri[:, 1], tri[:, 2]), side(tri[:, 2], tri[:, 0])
        cover = (~(((d1 < 0) | (d2 < 0) | (d3 < 0)) & ((d1 > 0) | (d2 > 0) | (d3 > 0)))).any(1).reshape(nx, ny)
        assert np.array_equal(cover, free), trial
        # largest component only: a subset of the free cells, connected
        big = sg.grid_to_rings(free, origin, cell, largest_only=True)
        inside_big = sg.rings_contain(big, X.ravel(), Y.ravel()).reshape(nx, ny)
        assert (inside_big <= free).all() and inside_big.any()
