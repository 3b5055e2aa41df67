#!/usr/bin/env python3
"""Pack the small, non-licensed DATA files the crowd_ppo path reads into one npz.

Runs only in the build container (reads /root/reference/motion/data).  The output
`egogen_amd/data/egogen_assets.npz` is data (index tables, one mocap seed, the room0
walkable polygon / navmesh / start-target pairs), no reference source text.

Sources (reference file -> key):
  data/SSM2.json                         -> marker_names, marker_ids   (main_ppo.py:296-300)
  data/CMU.json                          -> cmu_marker_ids             (exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py:117-118)
  data/smplx_vert_segmentation.json      -> feet_vids, part_names, vert_part (crowd_env_2f.py:53-59)
  data/locomotion/subseq_00343.npz       -> seed_*                     (environments.py:61-62,188-194)
  data/replica_room0_shapely.pkl         -> room0_ring_xy, room0_ring_off (environments.py:59-60)
  data/room0_samples.pkl                 -> room0_pairs                (environments.py:56-58)
  data/room_0/navmesh_tight.ply          -> room0_nav_v, room0_nav_f   (environments.py:54-55)
"""
import json
import os
import pickle
import struct
import sys

import numpy as np

REF = "/root/reference/motion/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "egogen_amd", "data", "egogen_assets.npz")


def parse_wkb_polygon(buf):
    """Minimal little-endian WKB Polygon reader -> list of [n,2] float64 rings."""
    assert buf[0] == 1
    (gtype,) = struct.unpack_from("<I", buf, 1)
    assert gtype == 3, gtype
    (nrings,) = struct.unpack_from("<I", buf, 5)
    off = 9
    rings = []
    for _ in range(nrings):
        (npts,) = struct.unpack_from("<I", buf, off)
        off += 4
        pts = np.frombuffer(buf, dtype="<f8", count=npts * 2, offset=off).reshape(npts, 2).copy()
        off += npts * 16
        rings.append(pts)
    return rings


class _WkbUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "shapely.io" and name == "from_wkb":
            return lambda b: parse_wkb_polygon(bytes(b))
        return super().find_class(module, name)


def parse_ply_binary(path):
    raw = open(path, "rb").read()
    hdr_end = raw.index(b"end_header\n") + len(b"end_header\n")
    hdr = raw[:hdr_end].decode().splitlines()
    nv = nf = 0
    for ln in hdr:
        if ln.startswith("element vertex"):
            nv = int(ln.split()[-1])
        if ln.startswith("element face"):
            nf = int(ln.split()[-1])
    off = hdr_end
    verts = np.zeros((nv, 3), np.float32)
    for i in range(nv):
        verts[i] = struct.unpack_from("<3f", raw, off)
        off += 12 + 4  # xyz + rgba
    faces = np.zeros((nf, 3), np.int32)
    for i in range(nf):
        n = raw[off]
        assert n == 3
        faces[i] = struct.unpack_from("<3i", raw, off + 1)
        off += 1 + 12
    return verts, faces


def main():
    out = {}
    ssm = json.load(open(f"{REF}/SSM2.json"))["markersets"][0]["indices"]
    out["marker_names"] = np.array(list(ssm.keys()))
    out["marker_ids"] = np.array(list(ssm.values()), np.int32)
    out["cmu_marker_ids"] = np.array(list(json.load(open(f"{REF}/CMU.json"))["markersets"][0]["indices"].values()), np.int32)
    seg = json.load(open(f"{REF}/smplx_vert_segmentation.json"))
    feet = []
    for part in ["leftToeBase", "rightToeBase", "leftFoot", "rightFoot"]:
        feet.extend(seg[part])
    out["feet_vids"] = np.array(sorted(set(feet)), np.int32)
    # per-vertex body-part id (first part listing the vertex); used only to lay out the SYNTHETIC template
    part_names = list(seg.keys())
    vert_part = np.full(10475, len(part_names) - 1, np.uint8)  # unlisted -> 'hips'
    seen = np.zeros(10475, bool)
    for i, n in enumerate(part_names):
        for v in seg[n]:
            if not seen[v]:
                seen[v] = True
                vert_part[v] = i
    out["part_names"] = np.array(part_names)
    out["vert_part"] = vert_part
    d = np.load(f"{REF}/locomotion/subseq_00343.npz", allow_pickle=True)
    out["seed_poses"] = d["poses"].astype(np.float64)
    out["seed_trans"] = d["trans"].astype(np.float64)
    out["seed_betas"] = d["betas"].astype(np.float64)
    out["seed_joints"] = d["joints"].astype(np.float32)
    out["seed_markers"] = d["marker_ssm2_67"].astype(np.float32)
    out["seed_transf_rotmat"] = d["transf_rotmat"]
    out["seed_transf_transl"] = d["transf_transl"]
    rings = _WkbUnpickler(open(f"{REF}/replica_room0_shapely.pkl", "rb")).load()
    out["room0_ring_xy"] = np.concatenate(rings, 0)
    out["room0_ring_off"] = np.cumsum([0] + [len(r) for r in rings]).astype(np.int32)
    pairs = pickle.load(open(f"{REF}/room0_samples.pkl", "rb"))
    out["room0_pairs"] = np.stack([np.concatenate([np.asarray(s).reshape(1, 3), np.asarray(t).reshape(1, 3)], 0)
                                   for s, t in pairs]).astype(np.float32)
    v, f = parse_ply_binary(f"{REF}/room_0/navmesh_tight.ply")
    out["room0_nav_v"] = v
    out["room0_nav_f"] = f
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
