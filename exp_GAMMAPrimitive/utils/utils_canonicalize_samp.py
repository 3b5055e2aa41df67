#!/usr/bin/env python3
"""Drop-in for the reference's exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py:
    python exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py 1      # 20-frame primitives  -> data/samp/Canonicalized-MP/
    python exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py 10     # 200-frame sequences  -> data/samp/Canonicalized-MPx10/
reads data/samp/<subset>*.pkl (SAMP `*_stageII.pkl`: pose_est_trans, pose_est_fullposes, shape_est_betas, mocap_framerate).
The body model is data/smplx/models/smplx/SMPLX_MALE.npz when present, else the synthetic one (say so with --num-verts)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from egogen_amd import setup_world as sw, synth  # noqa: E402
from egogen_amd.body_model import BodyModelHandle, SMPLXParser  # noqa: E402
from egogen_amd.canonicalize import canonicalize_samp  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("n_mps", type=int)
    ap.add_argument("--samp-path", default="data/samp")
    ap.add_argument("--num-verts", type=int, default=synth.NUM_VERTS)
    a = ap.parse_args(argv)
    bm, real = sw.load_body_model("male", num_verts=a.num_verts)
    if not real:
        print("data/smplx/models/smplx/SMPLX_MALE.npz not found: using the synthetic body model")
    h = BodyModelHandle(bm, synth.marker_ids(a.num_verts), synth.feet_vids(a.num_verts))
    parser = SMPLXParser({"n_batch": 20 * a.n_mps, "device": "cuda", "marker_placement": "ssm2_67", "body_models": {"male": h}})
    counts = canonicalize_samp(parser, a.n_mps, a.samp_path)
    print({k: v for k, v in counts.items() if v})


if __name__ == "__main__":
    main()
