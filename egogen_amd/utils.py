"""Host-side mirror of the reference's crowd_ppo/utils.py: `calc_sdf` (utils.py:54-84) on the HIP
kernel and `save_rollout_results` (utils.py:10-51) producing the same pickle layout."""
from __future__ import annotations

import ctypes as C
import os
import pickle
import time

import numpy as np
import torch

from . import _lib
from .body_model import SdfScene

_scene_cache = {}


def _scene_of(sdf_dict) -> SdfScene:
    if isinstance(sdf_dict, SdfScene):
        return sdf_dict
    key = id(sdf_dict["sdf"])
    sc = _scene_cache.get(key)
    if sc is None or sc._src is not sdf_dict["sdf"]:
        sc = SdfScene(sdf_dict)
        sc._src = sdf_dict["sdf"]
        _scene_cache.clear()
        _scene_cache[key] = sc
    return sc


def calc_sdf(vertices: torch.Tensor, sdf_dict, return_gradient: bool = False) -> torch.Tensor:
    """vertices[B,V,3] (cuda) , sdf_dict{'center','scale','sdf'} or SdfScene -> [B,V]; negative = penetrating."""
    if return_gradient:
        raise NotImplementedError("the reference's gradient branch is commented out (utils.py:69-81)")
    if vertices.dim() != 3 or vertices.shape[-1] != 3:
        raise ValueError("vertices must be [B,V,3]")
    sc = _scene_of(sdf_dict)
    B, V, _ = vertices.shape
    pts = vertices.to(dtype=torch.float32).contiguous()
    out = torch.empty(B, V, dtype=torch.float32, device=pts.device)
    rc = _lib.load().egx_sdf_sample(C.byref(sc.desc), _lib.ptr(pts), B * V, _lib.ptr(out), _lib.current_stream_ptr())
    _lib.check(rc, "egx_sdf_sample")
    return out


def save_rollout_results(scene, outmps, outfolder, man_id=None):
    """utils.py:10-51: one pickle per finished episode; plain dict / ndarray payload."""
    os.makedirs(outfolder, exist_ok=True)
    mp_keys = ["blended_marker", "smplx_params", "betas", "gender", "transf_rotmat", "transf_transl", "pelvis_loc", "mp_type"]
    to_np = lambda x: x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
    node = {"motion": [], "wpath": to_np(scene["wpath"]), "navmesh_path": scene.get("navmesh_path")}
    if "obj_id" in scene:
        node["obj_id"] = scene["obj_id"]
    if "obj_transform" in scene:
        node["obj_transform"] = (scene["obj_transform"],)
    if "scene_path" in scene:
        node["scene_path"] = scene["scene_path"]
    for mp in outmps:
        mp_node = {}
        for idx, key in enumerate(mp_keys):
            if key in ("gender", "mp_type", "betas", "transf_rotmat", "transf_transl"):
                mp_node[key] = mp[idx] if isinstance(mp[idx], str) else to_np(mp[idx])
            elif key == "smplx_params":
                mp_node[key] = to_np(mp[idx][0:1])
            else:
                mp_node[key] = to_np(mp[idx][0])
        node["motion"].append(mp_node)
    name = "motion_%s.pkl" % (str(time.time()) if man_id is None else man_id)
    path = os.path.join(outfolder, name)
    with open(path, "wb") as f:
        pickle.dump(node, f)
    return path


def rollout_primitives(motion_primitives, body_model, to_numpy: bool = True):
    """motion/vis.py:44-78 (also experiments/gen_egobody_depth.py:27-61): the `motion` list of a motion_*.pkl ->
    one continuous world-frame SMPL-X parameter sequence [t, 93].  Every primitive's 20 canonical-frame parameter rows
    are carried into the world frame of its (transf_rotmat, transf_transl) - transl' = R (transl + pelvis) - pelvis + T,
    glorot' = log(R exp(glorot)) - and the motion-seed frames a primitive shares with its predecessor (2 for '2-frame',
    1 for '1-frame') are dropped.  `body_model`: BodyModelHandle of the primitive's gender (the reference builds a smplx
    module for `pelvis_original`).  The frame change is egx_update_transl_glorot with the inverse frame
    (R^T, -R^T T): x' = R'^T (x + d - T') - d."""
    lib = _lib.load()
    if not motion_primitives:
        raise ValueError("empty motion list")
    dev = torch.device("cuda")
    chunks = []
    for idx, mp in enumerate(motion_primitives):
        xb = torch.as_tensor(np.asarray(mp["smplx_params"], np.float32)).reshape(-1, 93).to(dev).contiguous()
        n = xb.shape[0]
        betas = torch.as_tensor(np.asarray(mp["betas"], np.float32)).reshape(1, -1)[:, :10].to(dev).contiguous()
        R = torch.as_tensor(np.asarray(mp["transf_rotmat"], np.float32)).reshape(3, 3).to(dev)
        T = torch.as_tensor(np.asarray(mp["transf_transl"], np.float32)).reshape(3).to(dev)
        Rinv = R.t().contiguous().reshape(1, 3, 3)
        Tinv = (-(R.t() @ T)).reshape(1, 3).contiguous()
        zero = torch.zeros(1, 93, dtype=torch.float32, device=dev)
        delta = body_model.joints55(zero, betas, 1)[:, 0].repeat(n, 1).contiguous()      # pelvis of the rest pose
        out = torch.empty_like(xb)
        _lib.check(lib.egx_update_transl_glorot(_lib.ptr(Rinv), _lib.ptr(Tinv), 1, _lib.ptr(delta), _lib.ptr(xb), n, _lib.ptr(out),
                                                _lib.current_stream_ptr()), "egx_update_transl_glorot")
        start = 0 if idx == 0 else (2 if mp.get("mp_type") == "2-frame" else 1)
        chunks.append(out[start:])
    seq = torch.cat(chunks, dim=0)
    return seq.cpu().numpy() if to_numpy else seq
