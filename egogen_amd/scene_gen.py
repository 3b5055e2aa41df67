"""Scene preparation: what a new scene needs before `main_ppo.py` can train in it.

The reference ships its scene files ready-made - `data/room0_sdf.pkl` ({center, scale, sdf}: crowd_ppo/utils.py:54-84),
`room_0/navmesh_tight.ply`, `replica_room0_shapely.pkl`, `room0_samples.pkl` (environments.py:54-63) - and points to an external
tool for other scenes (README.md:97).  This module produces the same four artefacts from triangle meshes:

  signed-distance grid   `mesh_to_sdf_dict` / `scene_sdf_dict`  -> HIP kernel `egx_mesh_sdf` (one grid sample per lane)
  navmesh + polygon      `walkable_grid` -> `grid_to_navmesh`, `grid_to_rings`   (obstacle footprints inflated by the body radius
                         on a raster of the floor; rectangles of free cells as triangles, the free region's outline as rings)
  start / target pairs   `sample_pairs`
  files                  `write_ply` (binary little endian, readable by `egobody.read_ply` / trimesh), `save_scene` (npz)

The SDF needs the HIP library; everything else is host-side numpy (offline, once per scene)."""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------ meshes
def box_mesh(lo: Sequence[float], hi: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """Closed axis-aligned box: 8 vertices, 12 triangles, outward orientation."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    v = np.array([[(hi if (i >> a) & 1 else lo)[a] for a in range(3)] for i in range(8)], np.float64)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    f = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.int64)
    return v, f


def merge_meshes(meshes: Sequence[Tuple[np.ndarray, np.ndarray]]) -> Tuple[np.ndarray, np.ndarray]:
    vs, fs, off = [], [], 0
    for v, f in meshes:
        vs.append(np.asarray(v, np.float64))
        fs.append(np.asarray(f, np.int64) + off)
        off += len(v)
    return np.concatenate(vs, 0), np.concatenate(fs, 0)


def write_ply(path: str, vertices: np.ndarray, faces: np.ndarray) -> None:
    v, f = np.asarray(vertices, np.float32), np.asarray(faces, np.int32)
    with open(path, "wb") as fh:
        fh.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(v)}\nproperty float x\nproperty float y\n"
                  f"property float z\nelement face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        fh.write(v.astype("<f4").tobytes())
        for t in f:
            fh.write(struct.pack("<Biii", 3, int(t[0]), int(t[1]), int(t[2])))


# ------------------------------------------------------------------------------------------------ SDF
def mesh_to_sdf_dict(vertices: np.ndarray, faces: np.ndarray, res: int = 256, center: Optional[Sequence[float]] = None,
                     half: Optional[float] = None, inside_is_obstacle: bool = True, device: str = "cuda") -> Dict[str, "object"]:
    """Signed-distance grid of a closed mesh as an `sdf_dict` ({'center'[3], 'scale'[], 'sdf'[res,res,res]} float32 tensors on
    `device`, the layout of room0_sdf.pkl after crowd_env_2f.py:302-304).  The cube is center +- half (default: the mesh's
    bounding cube plus 10 %).  inside_is_obstacle: the stored value is > 0 inside the mesh (calc_sdf negates it), else < 0."""
    import torch
    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.EgxError("mesh_to_sdf_dict runs on the HIP device only (no CPU fallback)")
    v, f = np.asarray(vertices, np.float64), np.asarray(faces, np.int64)
    if center is None:
        center = (v.min(0) + v.max(0)) / 2
    if half is None:
        half = float(np.abs(v - np.asarray(center)).max()) * 1.1
    tris = torch.tensor(v[f].reshape(-1, 9), dtype=torch.float32, device=device).contiguous()
    grid = torch.empty(res, res, res, dtype=torch.float32, device=device)
    c = (C.c_float * 3)(*[float(x) for x in center])
    lib = _lib.load()
    _lib.check(lib.egx_mesh_sdf(_lib.ptr(tris), int(tris.shape[0]), c, float(1.0 / half), res, res, res, 1 if inside_is_obstacle else 0,
                                _lib.ptr(grid), _lib.current_stream_ptr()), "egx_mesh_sdf")
    return {"sdf": grid, "center": torch.tensor(np.asarray(center, np.float32), device=device),
            "scale": torch.tensor(np.float32(1.0 / half), device=device)}


def scene_sdf_dict(room: Tuple[np.ndarray, np.ndarray], obstacles: Optional[Tuple[np.ndarray, np.ndarray]], res: int = 256,
                   center: Sequence[float] = (0.0, 0.0, 1.0), half: float = 4.0, device: str = "cuda"):
    """Room shell (closed mesh whose interior is the free space) + obstacle solids -> one grid, < 0 in free space:
    max(d_room, d_obstacles) with d_room < 0 inside the room and d_obstacles > 0 inside an obstacle."""
    import torch
    d = mesh_to_sdf_dict(room[0], room[1], res, center, half, inside_is_obstacle=False, device=device)
    if obstacles is not None and len(obstacles[1]):
        o = mesh_to_sdf_dict(obstacles[0], obstacles[1], res, center, half, inside_is_obstacle=True, device=device)
        d["sdf"] = torch.maximum(d["sdf"], o["sdf"])
    return d


# ------------------------------------------------------------------------------------------------ walkable region
def _pt_tri_dist2_2d(px, py, tri):
    """Squared distance of points (px, py) [n] to a 2-D triangle [3,2] (0 inside)."""
    a, b, c = tri
    d2 = np.full(px.shape, np.inf)
    inside = np.ones(px.shape, bool)
    area = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    for p, q in ((a, b), (b, c), (c, a)):
        ex, ey = q[0] - p[0], q[1] - p[1]
        L = max(ex * ex + ey * ey, 1e-30)
        t = np.clip(((px - p[0]) * ex + (py - p[1]) * ey) / L, 0.0, 1.0)
        dx, dy = p[0] + t * ex - px, p[1] + t * ey - py
        d2 = np.minimum(d2, dx * dx + dy * dy)
        side = ex * (py - p[1]) - ey * (px - p[0])
        inside &= (side >= 0) if area >= 0 else (side <= 0)
    if abs(area) > 0:
        d2 = np.where(inside, 0.0, d2)
    return d2


def walkable_grid(floor_lo: Sequence[float], floor_hi: Sequence[float], obstacle_vertices: np.ndarray, obstacle_faces: np.ndarray,
                  radius: float = 0.2, cell: float = 0.05, z_range: Tuple[float, float] = (0.05, 2.0)):
    """Raster of the floor rectangle: a cell is free iff its centre is farther than `radius` from the footprint of every
    obstacle triangle that reaches into the height band `z_range` above the floor.  Returns (free[nx,ny] bool, origin[2], cell)."""
    lo, hi = np.asarray(floor_lo, np.float64)[:2], np.asarray(floor_hi, np.float64)[:2]
    nx, ny = int(round((hi[0] - lo[0]) / cell)), int(round((hi[1] - lo[1]) / cell))
    free = np.ones((nx, ny), bool)
    cx = lo[0] + (np.arange(nx) + 0.5) * cell
    cy = lo[1] + (np.arange(ny) + 0.5) * cell
    v, f = np.asarray(obstacle_vertices, np.float64), np.asarray(obstacle_faces, np.int64)
    for t in v[f] if len(f) else []:
        if t[:, 2].max() < z_range[0] or t[:, 2].min() > z_range[1]:
            continue
        i0 = max(int(np.floor((t[:, 0].min() - radius - lo[0]) / cell)), 0)
        i1 = min(int(np.ceil((t[:, 0].max() + radius - lo[0]) / cell)), nx)
        j0 = max(int(np.floor((t[:, 1].min() - radius - lo[1]) / cell)), 0)
        j1 = min(int(np.ceil((t[:, 1].max() + radius - lo[1]) / cell)), ny)
        if i0 >= i1 or j0 >= j1:
            continue
        X, Y = np.meshgrid(cx[i0:i1], cy[j0:j1], indexing="ij")
        d2 = _pt_tri_dist2_2d(X.ravel(), Y.ravel(), t[:, :2]).reshape(X.shape)
        free[i0:i1, j0:j1] &= d2 > radius * radius
    return free, lo.copy(), float(cell)


def grid_to_navmesh(free: np.ndarray, origin: np.ndarray, cell: float, floor_height: float = 0.0):
    """Free cells -> maximal rectangles (runs along y merged along x while identical) -> two triangles each.  Any triangle
    cover of the free region serves `get_map` (a point is walkable iff some triangle contains it, batch_gen_amass.py:949-961)."""
    nx, ny = free.shape
    rects, open_runs = [], {}
    for i in range(nx + 1):
        runs = set()
        if i < nx:
            col = np.concatenate([[False], free[i], [False]])
            edges = np.flatnonzero(col[1:] != col[:-1])
            runs = {(int(edges[k]), int(edges[k + 1])) for k in range(0, len(edges), 2)}
        for r in list(open_runs):
            if r not in runs:
                rects.append((open_runs.pop(r), i, r[0], r[1]))
        for r in runs:
            open_runs.setdefault(r, i)
    v, f = [], []
    for (i0, i1, j0, j1) in rects:
        x0, x1, y0, y1 = origin[0] + i0 * cell, origin[0] + i1 * cell, origin[1] + j0 * cell, origin[1] + j1 * cell
        b = len(v)
        v += [[x0, y0, floor_height], [x1, y0, floor_height], [x1, y1, floor_height], [x0, y1, floor_height]]
        f += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    return np.asarray(v, np.float64).reshape(-1, 3), np.asarray(f, np.int64).reshape(-1, 3)


def grid_to_rings(free: np.ndarray, origin: np.ndarray, cell: float, largest_only: bool = True) -> List[np.ndarray]:
    """Outline of the free region as closed rings [n,2] (collinear points removed, largest ring first).  With
    `largest_only`, cells not connected (4-neighbourhood) to the largest free component are dropped first - the
    reference keeps the biggest polygon of the navmesh union (environments.py:639-643)."""
    free = free.copy()
    if largest_only:
        lab = -np.ones(free.shape, np.int64)
        sizes = []
        for s in zip(*np.nonzero(free)):
            if lab[s] >= 0:
                continue
            stack, lab[s], cnt = [s], len(sizes), 0
            while stack:
                i, j = stack.pop()
                cnt += 1
                for a, b in ((i + 1, j), (i - 1, j), (i, j + 1), (i, j - 1)):
                    if 0 <= a < free.shape[0] and 0 <= b < free.shape[1] and free[a, b] and lab[a, b] < 0:
                        lab[a, b] = len(sizes)
                        stack.append((a, b))
            sizes.append(cnt)
        if sizes:
            free &= lab == int(np.argmax(sizes))
    P = np.pad(free, 1)
    nxt: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
    for i, j in zip(*np.nonzero(free)):
        pi, pj = i + 1, j + 1
        if not P[pi, pj - 1]:
            nxt.setdefault((i, j), []).append((i + 1, j))
        if not P[pi + 1, pj]:
            nxt.setdefault((i + 1, j), []).append((i + 1, j + 1))
        if not P[pi, pj + 1]:
            nxt.setdefault((i + 1, j + 1), []).append((i, j + 1))
        if not P[pi - 1, pj]:
            nxt.setdefault((i, j + 1), []).append((i, j))
    rings = []
    while nxt:
        start = next(iter(nxt))
        ring, cur = [start], start
        while True:
            outs = nxt[cur]
            step = outs.pop()
            if not outs:
                del nxt[cur]
            cur = step
            if cur == start:
                break
            ring.append(cur)
        pts = np.asarray(ring, np.float64)
        keep = []
        n = len(pts)
        for k in range(n):
            a, b, c = pts[k - 1], pts[k], pts[(k + 1) % n]
            if (b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0]) != 0:
                keep.append(k)
        pts = pts[keep]
        pts = np.concatenate([pts, pts[:1]], 0)
        rings.append(np.stack([origin[0] + pts[:, 0] * cell, origin[1] + pts[:, 1] * cell], 1))
    area = lambda r: abs(0.5 * np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))
    rings.sort(key=lambda r: -area(r))
    return rings


def rings_contain(rings: List[np.ndarray], x: np.ndarray, y: np.ndarray) -> np.ndarray:
    e = np.concatenate([np.concatenate([r[:-1], r[1:]], 1) for r in rings], 0)
    x, y = np.asarray(x, np.float64).reshape(-1), np.asarray(y, np.float64).reshape(-1)
    x0, y0, x1, y1 = (e[:, k][None, :] for k in range(4))
    with np.errstate(divide="ignore", invalid="ignore"):
        straddle = (y0 > y[:, None]) != (y1 > y[:, None])
        xint = x0 + (y[:, None] - y0) * (x1 - x0) / (y1 - y0)
    return (np.sum(straddle & (x[:, None] < xint), axis=1) & 1) == 1


def sample_pairs(rings: List[np.ndarray], n: int, min_dist: float = 1.7, seed: int = 0, floor_height: float = 0.0) -> np.ndarray:
    """n (start, target) pairs [n,2,3], both inside the walkable polygon and at least `min_dist` apart (the content of
    `*_samples.pkl`, environments.py:59-63)."""
    rng = np.random.default_rng(seed)
    lo = np.min([r.min(0) for r in rings], 0)
    hi = np.max([r.max(0) for r in rings], 0)
    out = np.zeros((0, 2, 3))
    while len(out) < n:
        p = rng.uniform(lo, hi, (4 * n, 2, 2))
        ok = rings_contain(rings, p[:, 0, 0], p[:, 0, 1]) & rings_contain(rings, p[:, 1, 0], p[:, 1, 1]) & \
            (np.linalg.norm(p[:, 0] - p[:, 1], axis=-1) >= min_dist)
        p = p[ok]
        out = np.concatenate([out, np.concatenate([p, np.full(p.shape[:2] + (1,), floor_height)], -1)], 0)
    return out[:n].astype(np.float32)


def box_scene_from_meshes(floor_lo, floor_hi, obstacle_mesh, radius: float = 0.2, cell: float = 0.05, n_pairs: int = 20000,
                          min_dist: float = 1.7, seed: int = 0, floor_height: float = 0.0) -> dict:
    """One entry of `VecCrowdEnv(scene_kind='box', box_scenes=[...])`: {'edges', 'tris', 'floor_height', 'pairs'} plus the
    navmesh ('nav_v', 'nav_f') and polygon ('rings') they were derived from."""
    from . import synth
    free, origin, cell = walkable_grid(floor_lo, floor_hi, obstacle_mesh[0], obstacle_mesh[1], radius, cell)
    nav_v, nav_f = grid_to_navmesh(free, origin, cell, floor_height)
    rings = grid_to_rings(free, origin, cell)
    return {"edges": synth.rings_to_edges(rings).astype(np.float32), "tris": nav_v[nav_f][:, :, :2].reshape(-1, 6).astype(np.float32),
            "floor_height": float(floor_height), "pairs": sample_pairs(rings, n_pairs, min_dist, seed, floor_height),
            "nav_v": nav_v, "nav_f": nav_f, "rings": rings}


def save_scene(path: str, scene: dict, sdf_dict: Optional[dict] = None) -> None:
    """npz pack of a generated scene (polygon rings flattened with offsets; the SDF grid if given)."""
    out = {"edges": scene["edges"], "tris": scene["tris"], "floor_height": np.float32(scene["floor_height"]), "pairs": scene["pairs"],
           "nav_v": scene["nav_v"], "nav_f": scene["nav_f"], "ring_xy": np.concatenate(scene["rings"], 0),
           "ring_off": np.cumsum([0] + [len(r) for r in scene["rings"]]).astype(np.int32)}
    if sdf_dict is not None:
        for k in ("sdf", "center", "scale"):
            v = sdf_dict[k]
            out["sdf_" + k] = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    np.savez_compressed(path, **out)
