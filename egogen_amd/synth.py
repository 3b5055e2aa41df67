"""Seeded synthetic assets for the crowd_ppo hot path (no licensed files needed).

The reference loads SMPLX_{MALE,FEMALE}.npz, VPoser, room0_sdf.pkl and the random-box scene
set (README.md:51-85 of the reference) - none of which can be shipped.  BASELINE.json asks for
"synthetic random-init bodies/scenes"; this module builds them deterministically (SURVEY.md
section 8(d)):

* `make_body_model`  - a body model with the exact SMPL-X tensor shapes (V=10475, 55 joints,
  486 pose-blend rows, 12 hand-PCA comps, 21 vertex joints, 51 face landmarks) laid out as a
  humanoid using the real per-vertex body-part table, so the real marker / feet index tables
  (SSM2.json, smplx_vert_segmentation.json) land on sensible places.
* `make_sdf_scene`   - analytic signed-distance grid of a room with one box (config 2).
* `make_box_scenes`  - floor + one random axis-aligned box: navmesh triangles, walkable polygon
  and start/target pairs (configs 3/4).

Everything is numpy on the host; callers upload what they need.
"""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ASSET_PACK = os.path.join(_HERE, "data", "egogen_assets.npz")

NUM_VERTS = 10475
NUM_JOINTS = 55
NUM_BODY_JOINTS = 21
NUM_BETAS = 10
NUM_PCA = 12
NUM_EXTRA_VJ = 21
NUM_LMK = 51
NUM_JOINTS_OUT = NUM_JOINTS + NUM_EXTRA_VJ + NUM_LMK  # 127
POSE_FEAT = (NUM_JOINTS - 1) * 9  # 486

# SMPL-X kinematic tree (55 joints): pelvis, legs/spine, arms, jaw+eyes, 15+15 finger joints.
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
     15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int32)

_assets_cache = None


def load_assets() -> Dict[str, np.ndarray]:
    """The packed DATA files of the reference (see scripts/make_data_pack.py)."""
    global _assets_cache
    if _assets_cache is None:
        with np.load(ASSET_PACK, allow_pickle=False) as z:
            _assets_cache = {k: z[k] for k in z.files}
    return _assets_cache


def _rest_joints() -> np.ndarray:
    """Approximate SMPL-X rest joints: x = body-left, y = up, z = forward (metres)."""
    J = np.zeros((NUM_JOINTS, 3), np.float64)
    body = {
        0: (0.00, -0.35, 0.00), 1: (0.06, -0.44, -0.01), 2: (-0.06, -0.44, -0.01), 3: (0.00, -0.24, -0.02),
        4: (0.11, -0.82, -0.02), 5: (-0.11, -0.82, -0.02), 6: (0.00, -0.10, 0.00),
        7: (0.09, -1.23, -0.06), 8: (-0.09, -1.23, -0.06), 9: (0.00, -0.05, 0.02),
        10: (0.12, -1.29, 0.07), 11: (-0.12, -1.29, 0.07), 12: (0.00, 0.17, -0.03),
        13: (0.05, 0.08, -0.01), 14: (-0.05, 0.08, -0.01), 15: (0.00, 0.26, 0.02),
        16: (0.18, 0.11, -0.02), 17: (-0.18, 0.11, -0.02), 18: (0.44, 0.10, -0.04), 19: (-0.44, 0.10, -0.04),
        20: (0.70, 0.10, -0.04), 21: (-0.70, 0.10, -0.04),
        22: (0.00, 0.24, 0.03), 23: (0.03, 0.31, 0.07), 24: (-0.03, 0.31, 0.07),
    }
    for j, p in body.items():
        J[j] = p
    # fingers: order index, middle, pinky, ring, thumb; 3 joints each
    spread = [0.02, 0.0, -0.04, -0.02, 0.04]
    for side, base, sgn in ((0, 25, 1.0), (1, 40, -1.0)):
        wrist = J[20 + side]
        for f in range(5):
            for k in range(3):
                j = base + 3 * f + k
                length = 0.09 + 0.03 * k if f != 4 else 0.03 + 0.03 * k
                J[j] = wrist + np.array([sgn * length, -0.005 * k, spread[f]])
    return J


# body part (smplx_vert_segmentation.json key) -> (joint_a, joint_b, radius): vertices of the part
# are scattered around the segment joint_a -> joint_b.
_PART_SEG = {
    "hips": (0, 3, 0.11), "spine": (3, 6, 0.11), "spine1": (6, 9, 0.11), "spine2": (9, 12, 0.12),
    "neck": (12, 15, 0.05), "head": (15, 15, 0.09), "leftEye": (23, 23, 0.012), "rightEye": (24, 24, 0.012),
    "eyeballs": (15, 15, 0.03),
    "leftShoulder": (13, 16, 0.06), "rightShoulder": (14, 17, 0.06),
    "leftArm": (16, 18, 0.045), "rightArm": (17, 19, 0.045),
    "leftForeArm": (18, 20, 0.035), "rightForeArm": (19, 21, 0.035),
    "leftHand": (20, 25, 0.03), "rightHand": (21, 40, 0.03),
    "leftHandIndex1": (-1, -1, 0.008), "rightHandIndex1": (-2, -2, 0.008),
    "leftUpLeg": (1, 4, 0.07), "rightUpLeg": (2, 5, 0.07),
    "leftLeg": (4, 7, 0.05), "rightLeg": (5, 8, 0.05),
    "leftFoot": (7, 10, 0.035), "rightFoot": (8, 11, 0.035),
    "leftToeBase": (10, 10, 0.03), "rightToeBase": (11, 11, 0.03),
}


def make_body_model(seed: int = 0, num_verts: int = NUM_VERTS, nnz_weights: int = 4, structured: bool = False) -> Dict[str, np.ndarray]:
    """Synthetic body model with SMPL-X shapes.  Keys mirror the fields smplx reads from
    SMPLX_*.npz after its own preprocessing (smplx/body_models.py [upstream, unpinned]):

      v_template[V,3] shapedirs[V,3,10] posedirs[486,3V] J_regressor[55,V] parents[55]
      lbs_weights[V,55] hand_comps_l/r[12,45] hand_mean_l/r[45]
      extra_vids[21] lmk_vids[51,3] lmk_bary[51,3]

    `num_verts` < 10475 gives a reduced model for fast unit tests (index tables are then
    remapped with `remap_ids`).

    `structured`: blend shapes with the STRUCTURE of a learned body model instead of i.i.d. noise (SURVEY 8(d)'s benchmark body,
    the default): shape directions are smooth fields over the body (a small random affine map of the rest position per
    component - limb lengths and girths - fading with the component index, plus 5 % detail) and pose correctives act near
    their joint (Gaussian fall-off of 12 cm).  Same shapes, same magnitudes at the affected vertices; what changes is that a
    vertex's offset from ITS joints is bounded by centimetres, which is what lets the work-item culling of the SDF path
    (csrc/body_model.hip) prove tiles free - with i.i.d. noise in every one of the 469 x 31 425 entries no a-priori bound
    is tighter than ~0.4 m.
    """
    rng = np.random.default_rng(seed)
    A = load_assets()
    V = num_verts
    J = _rest_joints()
    part_names = [str(x) for x in A["part_names"]]
    vert_part = A["vert_part"]
    if V != NUM_VERTS:
        vert_part = vert_part[np.linspace(0, NUM_VERTS - 1, V).astype(np.int64)]

    v_template = np.zeros((V, 3), np.float64)
    owner = np.zeros(V, np.int32)  # primary joint of each vertex
    for v in range(V):
        ja, jb, rad = _PART_SEG[part_names[vert_part[v]]]
        if ja < 0:  # finger vertices: spread over the 15 finger joints of that hand
            base = 25 if ja == -1 else 40
            jb = base + int(rng.integers(0, 15))
            ja = int(SMPLX_PARENTS[jb])
        t = rng.uniform(0.0, 1.0)
        p = J[ja] * (1 - t) + J[jb] * t
        v_template[v] = p + rng.normal(0.0, rad * 0.5, 3)
        owner[v] = jb if t > 0.5 else ja
    # feet: keep soles flat-ish and lowest (real feet index table lands here)
    for name, sole_y in (("leftFoot", -1.31), ("rightFoot", -1.31), ("leftToeBase", -1.31), ("rightToeBase", -1.31)):
        ids = np.where(vert_part == part_names.index(name))[0]
        v_template[ids, 1] = np.maximum(v_template[ids, 1], sole_y)
    body_lowest = v_template[:, 1].min()
    non_feet = ~np.isin(vert_part, [part_names.index(n) for n in ("leftFoot", "rightFoot", "leftToeBase", "rightToeBase")])
    v_template[non_feet, 1] = np.maximum(v_template[non_feet, 1], body_lowest + 0.04)

    shapedirs = rng.normal(0.0, 0.01, (V, 3, NUM_BETAS))
    posedirs = rng.normal(0.0, 0.002, (POSE_FEAT, V * 3))
    if structured:
        cen = v_template.mean(0)
        for k in range(NUM_BETAS):
            fade = 1.0 / (1.0 + k / 3.0)
            Bk = rng.normal(0.0, 0.03, (3, 3)) * fade
            ak = rng.normal(0.0, 0.01, 3) * fade
            shapedirs[:, :, k] = ak + (v_template - cen) @ Bk.T + 0.05 * shapedirs[:, :, k]
        pd = posedirs.reshape(NUM_JOINTS - 1, 9, V, 3)
        for j in range(1, NUM_JOINTS):
            fall = np.exp(-np.sum((v_template - J[j]) ** 2, axis=1) / (2 * 0.12 ** 2))       # [V]
            pd[j - 1] *= fall[None, :, None]
        posedirs = pd.reshape(POSE_FEAT, V * 3)

    # joint regressor: each row a normalised non-negative combination of 32 vertices owned by
    # (or nearest to) that joint, then shifted so J_regressor @ v_template == rest joint exactly-ish
    J_regressor = np.zeros((NUM_JOINTS, V), np.float64)
    for j in range(NUM_JOINTS):
        d = np.linalg.norm(v_template - J[j], axis=1)
        near = np.argsort(d)[:64]
        pick = rng.choice(near, size=min(32, V), replace=False)
        w = rng.uniform(0.2, 1.0, len(pick))
        J_regressor[j, pick] = w / w.sum()

    # skinning weights: nnz per vertex = owner, its parent, and nearest other joints (Dirichlet)
    lbs_weights = np.zeros((V, NUM_JOINTS), np.float64)
    Jd = np.linalg.norm(v_template[:, None, :] - J[None, :, :], axis=2)  # [V,55]
    for v in range(V):
        js = [int(owner[v])]
        par = int(SMPLX_PARENTS[owner[v]])
        if par >= 0:
            js.append(par)
        for cand in np.argsort(Jd[v]):
            if len(js) >= nnz_weights:
                break
            if int(cand) not in js:
                js.append(int(cand))
        w = rng.dirichlet(np.array([4.0] + [1.0] * (len(js) - 1)))
        lbs_weights[v, js] = w

    hand_comps_l = rng.normal(0.0, 0.1, (NUM_PCA, 45))
    hand_comps_r = rng.normal(0.0, 0.1, (NUM_PCA, 45))
    hand_mean_l = rng.normal(0.0, 0.05, 45)
    hand_mean_r = rng.normal(0.0, 0.05, 45)

    # 21 vertex-selected joints: nose, reye, leye, rear, lear, 6 feet, 10 finger tips
    def nearest(p):
        return int(np.argmin(np.linalg.norm(v_template - np.asarray(p), axis=1)))

    targets = [
        J[15] + [0, 0.02, 0.11],            # nose
        J[24] + [0, 0, 0.012],              # reye (surface, in front of eyeball joint 24)
        J[23] + [0, 0, 0.012],              # leye
        J[15] + [-0.08, 0.02, 0.0],         # rear
        J[15] + [0.08, 0.02, 0.0],          # lear
        J[10] + [0.02, -0.02, 0.05], J[10] + [0.05, -0.02, 0.02], J[7] + [0, -0.07, -0.04],   # LBigToe LSmallToe LHeel
        J[11] + [-0.02, -0.02, 0.05], J[11] + [-0.05, -0.02, 0.02], J[8] + [0, -0.07, -0.04],  # R...
    ]
    for base in (25, 40):
        for f in (4, 0, 1, 3, 2):  # thumb, index, middle, ring, pinky tips
            targets.append(J[base + 3 * f + 2] + [0.02 if base == 25 else -0.02, 0, 0])
    extra_vids = []
    used = set()
    for t in targets:
        order = np.argsort(np.linalg.norm(v_template - np.asarray(t), axis=1))
        for c in order:
            if int(c) not in used:
                used.add(int(c))
                extra_vids.append(int(c))
                break
    extra_vids = np.array(extra_vids, np.int32)
    # pin the eye-surface vertices exactly in front of the eyeball joints so the look-at vector
    # (crowd_env_2f.py:531) is well defined for the synthetic body
    v_template[extra_vids[1]] = J[24] + [0, 0, 0.012]
    v_template[extra_vids[2]] = J[23] + [0, 0, 0.012]

    head_ids = np.where(vert_part == part_names.index("head"))[0]
    if len(head_ids) < 3:
        head_ids = np.arange(V)
    lmk_vids = np.stack([rng.choice(head_ids, size=3, replace=False) for _ in range(NUM_LMK)]).astype(np.int32)
    lmk_bary = rng.dirichlet(np.ones(3), NUM_LMK)

    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {
        "v_template": f32(v_template), "shapedirs": f32(shapedirs), "posedirs": f32(posedirs),
        "J_regressor": f32(J_regressor), "parents": SMPLX_PARENTS.copy(), "lbs_weights": f32(lbs_weights),
        "hand_comps_l": f32(hand_comps_l), "hand_comps_r": f32(hand_comps_r),
        "hand_mean_l": f32(hand_mean_l), "hand_mean_r": f32(hand_mean_r),
        "extra_vids": extra_vids, "lmk_vids": lmk_vids, "lmk_bary": f32(lmk_bary),
    }


def remap_ids(ids: np.ndarray, num_verts: int) -> np.ndarray:
    """Map full-resolution vertex ids onto a reduced synthetic model (unit tests only)."""
    if num_verts == NUM_VERTS:
        return np.asarray(ids, np.int32)
    return ((np.asarray(ids, np.int64) * (num_verts - 1)) // (NUM_VERTS - 1)).astype(np.int32)


def marker_ids(num_verts: int = NUM_VERTS) -> np.ndarray:
    return remap_ids(load_assets()["marker_ids"], num_verts)


def feet_vids(num_verts: int = NUM_VERTS) -> np.ndarray:
    return np.unique(remap_ids(load_assets()["feet_vids"], num_verts))


def feet_marker_idx() -> List[int]:
    """main_ppo.py:298-299 of the reference: indices (into the 67 markers) of the 6 feet markers."""
    names = [str(x) for x in load_assets()["marker_names"]]
    return [names.index(n) for n in ["RHEE", "RTOE", "RRSTBEEF", "LHEE", "LTOE", "LRSTBEEF"]]


# ---------------------------------------------------------------------------------------------
# scenes
# ---------------------------------------------------------------------------------------------

def _box_sdf(p, lo, hi):
    """Signed distance to an axis-aligned box (negative inside).  p[...,3]."""
    c = (lo + hi) / 2
    h = (hi - lo) / 2
    q = np.abs(p - c) - h
    outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
    inside = np.minimum(np.max(q, axis=-1), 0.0)
    return outside + inside


def make_sdf_scene(res: int = 256, room: str = "single_box", seed: int = 0, obstacle=None) -> Dict[str, np.ndarray]:
    """Analytic SDF grid with the reference's storage convention (crowd_ppo/utils.py:54-84):
    grid value > 0 inside obstacles / outside the room, < 0 in free space, so that
    calc_sdf() = -trilinear(grid) is negative where a vertex penetrates.  Grid is indexed
    [x][y][z] over the cube `center +- 1/scale`.

    'single_box': 8 m cube centred (0,0,1); free space = room |x|,|y|<3.9, 0<z<4.9 minus a
    1x1x1 m box centred (1.5,0,0.5)   (SURVEY.md 8(d) config 2).
    'room0': same construction with the room0 footprint bounds (config 1).
    """
    center = np.array([0.0, 0.0, 1.0])
    half = 4.0
    lin = (np.arange(res, dtype=np.float64) + 0.5) / res * 2.0 - 1.0  # align_corners=False cell centres
    xs = center[0] + lin * half
    ys = center[1] + lin * half
    zs = center[2] + lin * half
    P = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1)
    if room == "room0":
        A = load_assets()
        lo_xy = A["room0_ring_xy"].min(0)
        hi_xy = A["room0_ring_xy"].max(0)
        room_lo = np.array([lo_xy[0], lo_xy[1], 0.0])
        room_hi = np.array([hi_xy[0], hi_xy[1], 4.9])
        obs_lo = np.array([(lo_xy[0] + hi_xy[0]) / 2 - 0.4, (lo_xy[1] + hi_xy[1]) / 2 - 0.4, 0.0])
        obs_hi = obs_lo + np.array([0.8, 0.8, 0.8])
    else:
        room_lo = np.array([-3.9, -3.9, 0.0])
        room_hi = np.array([3.9, 3.9, 4.9])
        obs_lo = np.array([1.0, -0.5, 0.0])
        obs_hi = np.array([2.0, 0.5, 1.0])
    if obstacle is not None:                    # (lo[3], hi[3]): the obstacle somewhere else (tests: a box on an agent)
        obs_lo, obs_hi = np.asarray(obstacle[0], np.float64), np.asarray(obstacle[1], np.float64)
    d_room = _box_sdf(P, room_lo, room_hi)      # <0 inside the room
    d_obs = _box_sdf(P, obs_lo, obs_hi)         # <0 inside the obstacle
    free = np.maximum(d_room, -d_obs)           # <0 in free space
    return {
        "sdf": np.ascontiguousarray(free, dtype=np.float32),
        "center": center.astype(np.float32),
        "scale": np.float32(1.0 / half),
        "room_lo": room_lo.astype(np.float32), "room_hi": room_hi.astype(np.float32),
        "obs_lo": obs_lo.astype(np.float32), "obs_hi": obs_hi.astype(np.float32),
    }


def rect_ring(lo, hi, ccw=True) -> np.ndarray:
    r = np.array([[lo[0], lo[1]], [hi[0], lo[1]], [hi[0], hi[1]], [lo[0], hi[1]], [lo[0], lo[1]]], np.float64)
    return r if ccw else r[::-1].copy()


def rings_to_edges(rings: List[np.ndarray]) -> np.ndarray:
    """Closed rings [n,2] (first == last) -> edge list [E,4] = (x0,y0,x1,y1)."""
    es = []
    for r in rings:
        r = np.asarray(r, np.float64)
        es.append(np.concatenate([r[:-1], r[1:]], 1))
    return np.concatenate(es, 0)


def sdf_scene_polygon(scene: Dict[str, np.ndarray]) -> List[np.ndarray]:
    """Walkable polygon (exterior + obstacle hole) matching `make_sdf_scene`."""
    return [rect_ring(scene["room_lo"][:2], scene["room_hi"][:2], True),
            rect_ring(scene["obs_lo"][:2], scene["obs_hi"][:2], False)]


def room0_polygon() -> List[np.ndarray]:
    A = load_assets()
    off = A["room0_ring_off"]
    return [A["room0_ring_xy"][off[i]:off[i + 1]].copy() for i in range(len(off) - 1)]


def box_scene_geometry(lo, hi):
    """8x8 m floor minus the axis-aligned hole [lo, hi] (xy): navmesh triangles [8,3,2] (4 rectangles) and the walkable
    polygon's rings (exterior ccw, hole cw)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    F_lo = np.array([-4.0, -4.0])
    F_hi = np.array([4.0, 4.0])
    rects = [(F_lo, np.array([F_hi[0], lo[1]])),
             (np.array([F_lo[0], hi[1]]), F_hi),
             (np.array([F_lo[0], lo[1]]), np.array([lo[0], hi[1]])),
             (np.array([hi[0], lo[1]]), np.array([F_hi[0], hi[1]]))]
    tris = []
    for a, b in rects:
        p00 = [a[0], a[1]]; p10 = [b[0], a[1]]; p11 = [b[0], b[1]]; p01 = [a[0], b[1]]
        tris.append([p00, p10, p11])
        tris.append([p00, p11, p01])
    return np.array(tris, np.float32), [rect_ring(F_lo, F_hi, True), rect_ring(lo, hi, False)]


def box_scene_from_hole(lo, hi, pairs=None) -> Dict[str, np.ndarray]:
    """One scene of the `make_box_scenes` form with the (inflated) obstacle footprint given (tests: a hole under an agent)."""
    tris, rings = box_scene_geometry(lo, hi)
    return {"tris": tris, "edges": rings_to_edges(rings).astype(np.float32),
            "pairs": np.zeros((1, 2, 3), np.float32) if pairs is None else np.asarray(pairs, np.float32),
            "box_lo": np.asarray(lo, np.float32), "box_hi": np.asarray(hi, np.float32), "floor_height": np.float32(0.0)}


def make_box_scenes(num_scenes: int = 64, pairs_per_scene: int = 2048, seed: int = 0) -> List[Dict[str, np.ndarray]]:
    """Synthetic stand-in for data/scenes/random_box_obstacle_new (environments.py:386-402):
    8x8 m floor, one axis-aligned box 0.5-1.5 m, navmesh = floor minus the box inflated by
    0.2 m (8 triangles), walkable polygon = same, start/target pairs >= 1.7 m apart in free space.
    """
    rng = np.random.default_rng(seed)
    scenes = []
    for s in range(num_scenes):
        size = rng.uniform(0.5, 1.5, 2)
        c = rng.uniform(-2.0, 2.0, 2)
        lo = c - size / 2 - 0.2
        hi = c + size / 2 + 0.2
        tris, rings = box_scene_geometry(lo, hi)
        pairs = np.zeros((pairs_per_scene, 2, 3), np.float32)
        n = 0
        while n < pairs_per_scene:
            cand = rng.uniform(-3.5, 3.5, (4 * pairs_per_scene, 2, 2))
            ok = np.ones(len(cand), bool)
            for k in range(2):
                inside = np.all((cand[:, k] > lo - 0.3) & (cand[:, k] < hi + 0.3), axis=1)
                ok &= ~inside
            ok &= np.linalg.norm(cand[:, 0] - cand[:, 1], axis=1) >= 1.7
            good = cand[ok][: pairs_per_scene - n]
            pairs[n:n + len(good), :, :2] = good
            n += len(good)
        scenes.append({"tris": tris, "edges": rings_to_edges(rings).astype(np.float32), "pairs": pairs,
                       "box_lo": lo.astype(np.float32), "box_hi": hi.astype(np.float32), "floor_height": np.float32(0.0)})
    return scenes


# ---------------------------------------------------------------------------------------------
# seeded network weights (bench / tests; the reference loads trained checkpoints instead)
# ---------------------------------------------------------------------------------------------

def seeded_fill(shapes: Dict[str, tuple], seed: int, bias_scale: float = 0.05, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Deterministic float32 tensors for a name->shape mapping (iteration order matters):
    matrices ~ N(0, gain^2/fan_in), vectors ~ N(0, bias_scale^2).  Used so that golden fixtures
    need to store only inputs/outputs, not multi-megabyte weight blobs."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in shapes.items():
        shp = tuple(int(s) for s in shp)
        if len(shp) >= 2:
            a = rng.standard_normal(shp) * (gain / np.sqrt(shp[-1]))
        elif name.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shp)
        elif name.endswith("num_batches_tracked"):
            a = np.zeros(shp)
        else:
            a = rng.standard_normal(shp) * bias_scale
        out[name] = a.astype(np.float32)
    return out
