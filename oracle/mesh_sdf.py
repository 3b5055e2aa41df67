"""TEST INFRASTRUCTURE - CPU check of the scene-preparation kernel `egx_mesh_sdf`.

PARITY UNPINNED: the reference has no SDF generator in its tree (it ships data/room0_sdf.pkl ready-made, README.md:97 points
to an external tool), so there is nothing to restate.  What IS fixed by the reference is the storage convention the grid must
follow - `calc_sdf` (crowd_ppo/utils.py:54-84): samples at the cell centres of `center +- 1/scale` (grid_sample,
align_corners=False), indexed [x][y][z], value > 0 inside obstacles because calc_sdf negates - and that is what this file
evaluates, with a formulation independent of the kernel's: distance = |plane distance| where the projection falls inside the
triangle, else the nearest of the three edge segments; sign from the generalised winding number (sum of the triangles' solid
angles, Van Oosterom & Strackee 1983) instead of a ray's crossing parity.  float64, O(samples x triangles): small grids only."""
import numpy as np


def sample_positions(center, scale, res):
    lin = (2 * np.arange(res, dtype=np.float64) + 1) / res - 1.0
    c = np.asarray(center, np.float64)
    X, Y, Z = np.meshgrid(c[0] + lin / scale, c[1] + lin / scale, c[2] + lin / scale, indexing="ij")
    return np.stack([X, Y, Z], -1)


def _seg_dist2(p, a, b):
    ab = b - a
    t = np.clip(((p - a) @ ab) / max(float(ab @ ab), 1e-300), 0.0, 1.0)
    d = a + t[:, None] * ab - p
    return np.einsum("ij,ij->i", d, d)


def mesh_signed_distance(vertices, faces, points, inside_positive=True):
    """points[n,3] -> signed distance[n] to the closed mesh."""
    v, f = np.asarray(vertices, np.float64), np.asarray(faces, np.int64)
    p = np.asarray(points, np.float64).reshape(-1, 3)
    best = np.full(len(p), np.inf)
    wind = np.zeros(len(p))
    for a, b, c in v[f]:
        n = np.cross(b - a, c - a)
        nn = float(n @ n)
        d2 = np.minimum(np.minimum(_seg_dist2(p, a, b), _seg_dist2(p, b, c)), _seg_dist2(p, c, a))
        if nn > 0:
            h = (p - a) @ n / nn                              # signed plane distance / |n|
            q = p - h[:, None] * n                            # projection
            inside = np.ones(len(p), bool)
            for s, e in ((a, b), (b, c), (c, a)):
                inside &= np.cross(e - s, q - s) @ n >= 0
            d2 = np.where(inside, h * h * nn, d2)
        best = np.minimum(best, d2)
        ra, rb, rc = a - p, b - p, c - p
        la, lb, lc = (np.linalg.norm(r, axis=1) for r in (ra, rb, rc))
        num = np.einsum("ij,ij->i", ra, np.cross(rb, rc))
        den = la * lb * lc + np.einsum("ij,ij->i", ra, rb) * lc + np.einsum("ij,ij->i", ra, rc) * lb + np.einsum("ij,ij->i", rb, rc) * la
        wind += 2 * np.arctan2(num, den)
    inside = np.abs(wind) / (4 * np.pi) > 0.5
    d = np.sqrt(best)
    return np.where(inside == bool(inside_positive), d, -d)
