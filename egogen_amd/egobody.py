"""Host side of the EgoBody evaluation (reference: exp_GAMMAPrimitive/utils/environments.py:630-783 `Egobody`,
crowd_ppo/main_egobody_eval.py): scene files -> walkable polygon, start / target sampling for two people who swap places,
random gender / body shape / motion seed per scene.  Everything per step runs in the crowd kernels (crowd_env.CrowdGroupEnv).

trimesh / shapely are not needed: the navmesh is read with `read_ply`, the union of its triangles
(`union_all(walkable_region)`, environments.py:633-638) is the set of its boundary edges chained into rings
(`navmesh_walkable_rings`; for the in-tree room_0 navmesh this reproduces the reference's own `replica_room0_shapely.pkl`
polygon, tests/test_egobody_cpu.py)."""
from __future__ import annotations

import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np

_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_ply(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Vertices [V,3] float64 and triangle faces [F,3] int64 of an ascii / binary_little_endian PLY (what
    `trimesh.load(path, force='mesh')` yields for the navmesh files: `navmesh_tight.ply`, environments.py:632)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    fmt, elements = None, []
    for ln in data[:end].decode("ascii", "replace").splitlines():
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
        elif tok[0] == "property":
            elements[-1]["props"].append(tok[1:])
    if fmt not in ("ascii", "binary_little_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
    verts, faces = None, []
    if fmt == "ascii":
        lines = data[end:].decode("ascii").split("\n")
        li = 0
        for el in elements:
            rows = [lines[li + i].split() for i in range(el["count"])]
            li += el["count"]
            if el["name"] == "vertex":
                names = [p[-1] for p in el["props"]]
                ix = [names.index(c) for c in ("x", "y", "z")]
                verts = np.array([[float(r[i]) for i in ix] for r in rows], np.float64).reshape(-1, 3)
            elif el["name"] == "face":
                for r in rows:
                    n = int(r[0])
                    idx = [int(v) for v in r[1:1 + n]]
                    faces += [[idx[0], idx[i], idx[i + 1]] for i in range(1, n - 1)]
    else:
        off = end
        for el in elements:
            if el["name"] == "vertex":
                names = [p[-1] for p in el["props"]]
                rec = "<" + "".join(_PLY_TYPES[p[0]] for p in el["props"])
                size = struct.calcsize(rec)
                ix = [names.index(c) for c in ("x", "y", "z")]
                rows = [struct.unpack_from(rec, data, off + i * size) for i in range(el["count"])]
                verts = np.array([[r[i] for i in ix] for r in rows], np.float64).reshape(-1, 3)
                off += size * el["count"]
            else:
                for _ in range(el["count"]):
                    row = None
                    for p in el["props"]:
                        if p[0] == "list":
                            n = struct.unpack_from("<" + _PLY_TYPES[p[1]], data, off)[0]
                            off += struct.calcsize(_PLY_TYPES[p[1]])
                            vals = struct.unpack_from("<" + _PLY_TYPES[p[2]] * n, data, off)
                            off += struct.calcsize(_PLY_TYPES[p[2]]) * n
                            if p[-1] in ("vertex_indices", "vertex_index"):
                                row = list(vals)
                        else:
                            off += struct.calcsize(_PLY_TYPES[p[0]])
                    if el["name"] == "face" and row is not None:
                        faces += [[row[0], row[i], row[i + 1]] for i in range(1, len(row) - 1)]
    if verts is None:
        raise ValueError(f"{path}: no vertex element")
    return verts, np.asarray(faces, np.int64).reshape(-1, 3)


def _ring_area(r: np.ndarray) -> float:
    x, y = r[:, 0], r[:, 1]
    return 0.5 * float(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))


def navmesh_walkable_rings(vertices: np.ndarray, faces: np.ndarray, decimals: int = 9) -> List[np.ndarray]:
    """Rings (closed, [n,2], exterior first) of the largest connected region of the union of the navmesh triangles in the xy
    plane - `union_all([Polygon(tri) ...])`, biggest area kept if the union is a MultiPolygon (environments.py:633-643).
    For a conforming triangulation the union's boundary is the set of edges that belong to exactly one triangle."""
    v2 = np.round(np.asarray(vertices, np.float64)[:, :2], decimals)
    uniq, inv = np.unique(v2, axis=0, return_inverse=True)   # navmesh files repeat vertices at the same position
    inv = inv.reshape(-1)
    f = inv[np.asarray(faces, np.int64)]
    f = f[(f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])]
    # connected components over shared edges
    parent = list(range(len(f)))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    owner: Dict[Tuple[int, int], List[int]] = {}
    for fi, t in enumerate(f):
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            owner.setdefault((min(a, b), max(a, b)), []).append(fi)
    for fs in owner.values():
        for o in fs[1:]:
            parent[find(o)] = find(fs[0])
    comp = np.array([find(i) for i in range(len(f))])
    tri = uniq[f]
    tri_area = 0.5 * np.abs((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1]) -
                            (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1]))
    best = max(set(comp.tolist()), key=lambda c: float(tri_area[comp == c].sum()))
    # boundary edges of that component, chained into rings
    nxt: Dict[int, List[int]] = {}
    for (a, b), fs in owner.items():
        if len(fs) == 1 and comp[fs[0]] == best:
            nxt.setdefault(a, []).append(b)
            nxt.setdefault(b, []).append(a)
    rings, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        ring, prev, cur = [start], None, start
        seen.add(start)
        while True:
            cand = [n for n in nxt[cur] if n != prev]
            step = next((n for n in cand if n not in seen), None)
            if step is None:
                break
            ring.append(step)
            seen.add(step)
            prev, cur = cur, step
        ring.append(start)
        rings.append(uniq[np.asarray(ring)])
    rings.sort(key=lambda r: -abs(_ring_area(r)))
    return rings


def _in_rings(edges: np.ndarray, x: float, y: float) -> bool:
    x0, y0, x1, y1 = edges[:, 0], edges[:, 1], edges[:, 2], edges[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        straddle = (y0 > y) != (y1 > y)
        xint = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
    return (int(np.sum(straddle & (x < xint))) & 1) == 1


def _dist_to_edges(edges: np.ndarray, x: float, y: float) -> float:
    a, d = edges[:, :2], edges[:, 2:] - edges[:, :2]
    t = np.clip(((x - a[:, 0]) * d[:, 0] + (y - a[:, 1]) * d[:, 1]) / np.maximum((d * d).sum(1), 1e-30), 0.0, 1.0)
    px, py = a[:, 0] + t * d[:, 0] - x, a[:, 1] + t * d[:, 1] - y
    return float(np.sqrt(px * px + py * py).min())


class EgobodySampler:
    """`Egobody.next_body` (environments.py:768-783): start and target are drawn on the navmesh surface until a 0.3 m disc
    around each lies in the walkable region and they are 1.5-5 m apart; the second person walks the opposite way; one random
    gender for both; per person a random motion-seed file and start frame and a random shape betas ~ N(0, 0.3^2)
    (gen_init_body, :679-700).  The rigid placement (face the target, +-0.2*2pi yaw, feet on the floor, :702-747) is done by
    the crowd reset kernel.  `Point.buffer(0.3)` is a 64-gon in shapely; the exact disc is used here (stricter by < 0.4 mm)."""

    def __init__(self, navmesh_vertices: np.ndarray, navmesh_faces: np.ndarray, motion_seeds: Sequence[dict], seed: int = 0,
                 scene_path: str = "mesh_floor_zup.ply", navmesh_path: str = "navmesh_tight.ply"):
        self.v = np.asarray(navmesh_vertices, np.float64)
        self.f = np.asarray(navmesh_faces, np.int64)
        self.rings = navmesh_walkable_rings(self.v, self.f)
        self.edges = np.concatenate([np.concatenate([r[:-1], r[1:]], 1) for r in self.rings], 0)
        self.motion_seeds = list(motion_seeds)
        if not self.motion_seeds:
            raise ValueError("EgobodySampler needs at least one motion seed {poses[n,>=66], trans[n,3]}")
        self.rng = np.random.default_rng(seed)
        tri = self.v[self.f]
        self._area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        self._cdf = np.cumsum(self._area) / self._area.sum()
        self.scene_path, self.navmesh_path = scene_path, navmesh_path

    @classmethod
    def from_scene_dir(cls, scene_dir: str, motion_seeds, seed: int = 0):
        import os
        v, f = read_ply(os.path.join(scene_dir, "navmesh_tight.ply"))
        return cls(v, f, motion_seeds, seed, os.path.join(scene_dir, "mesh_floor_zup.ply"), os.path.join(scene_dir, "navmesh_tight.ply"))

    def _surface_point(self) -> np.ndarray:
        t = self.v[self.f[int(np.searchsorted(self._cdf, self.rng.random()))]]
        u, w = self.rng.random(), self.rng.random()
        if u + w > 1.0:
            u, w = 1.0 - u, 1.0 - w
        return t[0] + u * (t[1] - t[0]) + w * (t[2] - t[0])

    def _free_point(self) -> np.ndarray:
        for _ in range(100000):
            p = self._surface_point()
            if _in_rings(self.edges, p[0], p[1]) and _dist_to_edges(self.edges, p[0], p[1]) >= 0.3:
                return p
        raise RuntimeError("no navmesh point has 0.3 m of clearance")

    def _seed(self) -> dict:
        ms = self.motion_seeds[int(self.rng.integers(len(self.motion_seeds)))]
        poses, trans = np.asarray(ms["poses"], np.float64), np.asarray(ms["trans"], np.float64)
        s = int(self.rng.integers(0, len(poses) - 1))
        return {"poses": poses[s:s + 2, :66], "trans": trans[s:s + 2], "betas": self.rng.normal(0.0, 0.3, 10)}

    def next_body(self) -> Tuple[dict, dict]:
        while True:
            start, target = self._free_point(), self._free_point()
            if 1.5 <= np.linalg.norm(target - start) <= 5.0:
                break
        gender = "male" if self.rng.random() < 0.5 else "female"
        out = []
        for a, b in ((start, target), (target, start)):
            out.append({"wpath": np.stack([a, b]).astype(np.float32), "gender": gender, "seed": self._seed(),
                        "scene_path": self.scene_path, "navmesh_path": self.navmesh_path, "floor_height": 0})
        return out[0], out[1]
