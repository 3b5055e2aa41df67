"""SAMP / AMASS sequences -> canonicalised motion primitives (reference:
exp_GAMMAPrimitive/utils/utils_canonicalize_samp.py, run as `utils_canonicalize_samp.py 1` and `... 10`, README.md:88-91).

Every sub-sequence of 20 N frames (120 fps mocap down-sampled by 3) is moved into the frame of its first body - origin at
the pelvis, x from the left to the right hip projected on the floor, z up (:60-86) - with the pelvis-offset correction of
:28-55 for the translation, and the 22 joints and the CMU-41 / SSM2-67 marker trajectories are extracted in that frame
(:150-189, main loop :192-290).  The frame change and the body model run on the GPU through `SMPLXParser`
(egx_lbs_joints / egx_canonical_frame / egx_update_transl_glorot / egx_lbs_forward); file handling is host-side."""
from __future__ import annotations

import glob
import os
import pickle
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import synth
from .body_model import SMPLXParser

SUBSETS = ("chair", "armchair", "highstool", "lie_down", "locomotion", "reebokstep", "run", "sofa", "table")   # :204
MP_FRAME, DOWNSAMPLE = 20, 3


def canonicalize_frames(parser: SMPLXParser, transl: np.ndarray, pose: np.ndarray, betas: np.ndarray, gender: str = "male",
                        cmu_marker_ids: Optional[Sequence[int]] = None, fps: float = 120.0) -> Dict[str, np.ndarray]:
    """One sub-sequence (already cut and down-sampled): transl [T,3], pose [T,>=66] (global_orient | body_pose | ...), betas
    [>=10] -> the dict utils_canonicalize_samp.py:240-283 saves (same keys, dtypes and shapes)."""
    T = transl.shape[0]
    pose = np.array(pose, np.float64, copy=True)
    b10 = np.asarray(betas, np.float64)[:10]
    xb = torch.zeros(T, 93, dtype=torch.float32, device="cuda")
    xb[:, :3] = torch.as_tensor(transl, dtype=torch.float32)
    xb[:, 3:69] = torch.as_tensor(pose[:, :66], dtype=torch.float32)
    bt = torch.as_tensor(b10, dtype=torch.float32).reshape(1, 10).cuda()
    R, Tr = parser.get_new_coordinate(bt, gender, xb[:1], to_numpy=False)          # frame of the first body (:251)
    new = parser.update_transl_glorot(R, Tr, bt, gender, xb, to_numpy=False, inplace=False)   # :253-259
    out = parser._bm(gender).forward(new, bt, T, want_verts=True, want_joints=True, want_markers=True)
    verts = out["vertices"]
    cmu = synth.remap_ids(synth.load_assets()["cmu_marker_ids"], verts.shape[1]) if cmu_marker_ids is None else np.asarray(cmu_marker_ids)
    pose[:, :3] = new[:, 3:6].double().cpu().numpy()
    return {"transf_rotmat": R[0].double().cpu().numpy(), "transf_transl": Tr[0].cpu().numpy().astype(np.float32),
            "trans": new[:, :3].double().cpu().numpy(), "poses": pose, "betas": b10, "gender": np.asarray(gender),
            "mocap_framerate": np.int64(int(fps)), "joints": out["joints"][:, :22].cpu().numpy().astype(np.float32),
            "marker_cmu_41": verts[:, torch.as_tensor(cmu, dtype=torch.long, device=verts.device)].cpu().numpy().astype(np.float32),
            "marker_ssm2_67": out["markers"].cpu().numpy().astype(np.float32)}


def canonicalize_subsequence(parser: SMPLXParser, seq: str, start_frame: int, end_frame: int):
    """utils_canonicalize_samp.py:123-189: frames [start, end) of one SAMP pickle, every third frame."""
    with open(seq, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    assert data["mocap_framerate"] == 120.0
    if data["pose_est_trans"].shape[0] <= end_frame:
        return None
    return canonicalize_frames(parser, data["pose_est_trans"][start_frame:end_frame:DOWNSAMPLE],
                               data["pose_est_fullposes"][start_frame:end_frame:DOWNSAMPLE], data["shape_est_betas"], "male",
                               fps=data["mocap_framerate"])


def canonicalize_samp(parser: SMPLXParser, n_mps: int, samp_dataset_path: str = "data/samp", subsets: Sequence[str] = SUBSETS,
                      verbose: bool = True) -> Dict[str, int]:
    """The `__main__` loop (:192-290): data/samp/<subset>*.pkl -> data/samp/Canonicalized-MP[xN]/data/<subset>/subseq_%05d.npz;
    sub-sequences are cut back to back (t += 20 N), too-short sequences and tails are skipped, files are indexed per subset."""
    len_subseq = MP_FRAME * int(n_mps)
    out_root = os.path.join(samp_dataset_path, f"Canonicalized-MPx{n_mps:d}" if n_mps > 1 else "Canonicalized-MP", "data")
    counts = {}
    for subset in subsets:
        seqs = glob.glob(os.path.join(samp_dataset_path, f"{subset}*.pkl"))
        outfolder = os.path.join(out_root, subset)
        os.makedirs(outfolder, exist_ok=True)
        if verbose:
            print(f"-- processing subset {subset:s}")
        index = 0
        for seq in seqs:
            with open(seq, "rb") as f:
                data = pickle.load(f, encoding="latin1")
            assert data["mocap_framerate"] == 120.0
            transl_all = data["pose_est_trans"][::DOWNSAMPLE]
            pose_all = data["pose_est_fullposes"][::DOWNSAMPLE]
            n = transl_all.shape[0]
            if n < len_subseq:
                continue
            for t in range(0, n - len_subseq + 1, len_subseq):
                d = canonicalize_frames(parser, transl_all[t:t + len_subseq], pose_all[t:t + len_subseq], data["shape_est_betas"][:10],
                                        "male", fps=data["mocap_framerate"])
                np.savez(os.path.join(outfolder, f"subseq_{index:05d}.npz"), **d)
                index += 1
        counts[subset] = index
    return counts
