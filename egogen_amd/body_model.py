"""Body-model operator: host-side mirror of the reference's `SMPLXParser` (models/baseops.py:271-598)
on top of the HIP library.  Same method names, argument meaning and output shapes; tensors stay on the
GPU (`to_numpy=True` copies back like the reference does).

Differences that are extensions, not behaviour changes:
  * `betas` may be [A,10] (one shape per agent, B = A * frames) instead of a single (10,) vector;
  * `forward_lbs(...)` exposes the fused outputs (joints + markers + SDF penetration counts) the
    crowd_ppo step needs without materialising the [B,V,3] vertex tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .synth import NUM_JOINTS_OUT


def _as_f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _as_i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


class SdfScene:
    """Device copy of the reference's `sdf_dict` {'center','scale','sdf'} (crowd_ppo/utils.py:54-58)."""

    def __init__(self, sdf_dict: Dict, device="cuda"):
        g = sdf_dict["sdf"]
        g = g if isinstance(g, torch.Tensor) else torch.as_tensor(np.asarray(g))
        g = g.squeeze()
        if g.dim() != 3:
            raise ValueError("sdf grid must be 3-D")
        self.grid = g.to(device=device, dtype=torch.float32).contiguous()
        c = sdf_dict["center"]
        c = c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c)
        s = sdf_dict["scale"]
        s = float(s.item() if hasattr(s, "item") else s)
        self.desc = _lib.SdfGrid()
        self.desc.grid = self.grid.data_ptr()
        self.desc.d0, self.desc.d1, self.desc.d2 = [int(x) for x in self.grid.shape]
        c = c.reshape(-1).astype(np.float32)
        self.desc.center[0], self.desc.center[1], self.desc.center[2] = float(c[0]), float(c[1]), float(c[2])
        self.desc.scale = s
        self.desc.coarse_minmax = None
        if self.grid.is_cuda:
            lib = _lib.load()
            nbytes = lib.egx_sdf_coarse_bytes(self.desc.d0, self.desc.d1, self.desc.d2)
            self.coarse = torch.empty(nbytes, dtype=torch.uint8, device=self.grid.device)
            _lib.check(lib.egx_sdf_build_coarse(C.byref(self.desc), _lib.ptr(self.coarse), _lib.current_stream_ptr()),
                       "egx_sdf_build_coarse")
            self.desc.coarse_minmax = self.coarse.data_ptr()


class BodyModelHandle:
    """One immutable device-resident body model (per gender).  Thread-safe for concurrent forwards on
    distinct streams as long as each caller passes its own workspace."""

    def __init__(self, bm: Dict[str, np.ndarray], marker_vids, feet_vids=()):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.EgxError("no HIP device visible - the body model operator has no CPU fallback")
        self._keep = {k: _as_f32(bm[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights",
                                                   "hand_comps_l", "hand_comps_r", "hand_mean_l", "hand_mean_r", "lmk_bary")}
        self._keep.update({k: _as_i32(bm[k]) for k in ("parents", "extra_vids", "lmk_vids")})
        self._keep["marker"] = _as_i32(marker_vids)
        self._keep["feet"] = _as_i32(feet_vids)
        k = self._keep
        d = _lib.BodyModelHost()
        d.num_verts = int(k["v_template"].shape[0])
        assert k["shapedirs"].shape == (d.num_verts, 3, 10), k["shapedirs"].shape
        assert k["posedirs"].shape == (486, 3 * d.num_verts), k["posedirs"].shape
        for f, key in (("v_template_host", "v_template"), ("shapedirs_host", "shapedirs"), ("posedirs_host", "posedirs"),
                       ("J_regressor_host", "J_regressor"), ("parents_host", "parents"), ("lbs_weights_host", "lbs_weights"),
                       ("hand_comps_l_host", "hand_comps_l"), ("hand_comps_r_host", "hand_comps_r"),
                       ("hand_mean_l_host", "hand_mean_l"), ("hand_mean_r_host", "hand_mean_r"),
                       ("extra_vids_host", "extra_vids"), ("lmk_vids_host", "lmk_vids"), ("lmk_bary_host", "lmk_bary"),
                       ("marker_vids_host", "marker"), ("feet_vids_host", "feet")):
            setattr(d, f, k[key].ctypes.data)
        d.num_markers = int(k["marker"].shape[0])
        d.num_feet = int(k["feet"].shape[0])
        h = C.c_void_p()
        _lib.check(lib.egx_body_model_create(C.byref(d), C.byref(h)), "egx_body_model_create")
        self.handle = h
        self.V = d.num_verts
        self.M = d.num_markers
        self.marker_vids = [int(v) for v in k["marker"]]
        self.nnz = lib.egx_body_model_nnz(h)
        # vertices a call without vertex output evaluates (tiles without picks / counted vertices are skipped)
        self.lbs_vertices = {"picks": lib.egx_body_model_lbs_vertices(h, 0), "sdf": lib.egx_body_model_lbs_vertices(h, 1)}
        mg = C.c_float()
        # SDF launches skip work items that are provably in free space when the model's blend shapes allow a tight bound
        self.culls = bool(lib.egx_body_model_culls(h, C.byref(mg)))
        self.cull_reference_margin = float(mg.value)
        self._ws: Dict[tuple, torch.Tensor] = {}
        del self._keep  # device copies are owned by the handle now

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                _lib.load().egx_body_model_destroy(h)
            except Exception:
                pass
            self.handle = None

    def workspace(self, B: int) -> torch.Tensor:
        """Scratch of one forward, cached per (batch size, stream): forwards on distinct streams never share it."""
        key = (B, torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = _lib.load().egx_lbs_workspace_bytes(self.handle, B)
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            self._ws[key] = ws
        return ws

    def cull_stats(self, B: int):
        """(items evaluated, items of an unculled call) of the LAST culled SDF forward of B bodies on the current stream's
        workspace (`egx_lbs_cull_stats`; synchronises with the host).  An item = one vertex tile x 256 bodies."""
        lib = _lib.load()
        act, tot = C.c_int32(), C.c_int32()
        torch.cuda.current_stream().synchronize()
        _lib.check(lib.egx_lbs_cull_stats(self.handle, _lib.ptr(self.workspace(B)), int(B), C.byref(act), C.byref(tot)), "egx_lbs_cull_stats")
        return int(act.value), int(tot.value)

    def fix_stats(self, B: int) -> int:
        """Vertices the LAST SDF-counting forward of B bodies on the current stream's workspace re-evaluated in fp32 (blend mode
        "f16mix": the ones its fp16 product could not decide; `egx_lbs_fix_stats`, synchronises with the host)."""
        lib = _lib.load()
        n = C.c_int32()
        torch.cuda.current_stream().synchronize()
        _lib.check(lib.egx_lbs_fix_stats(self.handle, _lib.ptr(self.workspace(B)), int(B), C.byref(n)), "egx_lbs_fix_stats")
        return int(n.value)

    def forward(self, xb: torch.Tensor, betas: torch.Tensor, frames_per_agent: int, want_verts=False,
                want_joints=True, want_markers=True, sdf: Optional[SdfScene] = None,
                R0: Optional[torch.Tensor] = None, T0: Optional[torch.Tensor] = None, out: Optional[dict] = None):
        """xb[B,93], betas[A,10] (A*frames_per_agent == B).  Returns dict with the requested outputs."""
        lib = _lib.load()
        B = int(xb.shape[0])
        if xb.dim() != 2 or xb.shape[1] != 93:
            raise ValueError(f"xb must be [B,93], got {tuple(xb.shape)}")
        if B == 0:
            raise ValueError("empty batch")
        A = int(betas.shape[0])
        if betas.dim() != 2 or betas.shape[1] != 10 or A * frames_per_agent != B:
            raise ValueError(f"betas must be [B/frames_per_agent,10]; got {tuple(betas.shape)} for B={B}, fpa={frames_per_agent}")
        xb = xb.to(dtype=torch.float32).contiguous()
        betas = betas.to(dtype=torch.float32).contiguous()
        out = out if out is not None else {}
        dev = xb.device
        if want_verts and "vertices" not in out:
            out["vertices"] = torch.empty(B, self.V, 3, dtype=torch.float32, device=dev)
        if want_joints and "joints" not in out:
            out["joints"] = torch.empty(B, NUM_JOINTS_OUT, 3, dtype=torch.float32, device=dev)
        if want_markers and "markers" not in out:
            out["markers"] = torch.empty(B, self.M, 3, dtype=torch.float32, device=dev)
        if sdf is not None and "pene_count" not in out:
            out["pene_count"] = torch.empty(B, dtype=torch.int32, device=dev)
        if R0 is not None:
            R0 = R0.to(torch.float32).reshape(A, 9).contiguous()
            T0 = T0.to(torch.float32).reshape(A, 3).contiguous()
        ws = self.workspace(B)
        rc = lib.egx_lbs_forward(self.handle, _lib.ptr(xb), _lib.ptr(betas), B, int(frames_per_agent),
                                 _lib.ptr(out.get("vertices")) if want_verts else None,
                                 _lib.ptr(out.get("joints")) if want_joints else None,
                                 _lib.ptr(out.get("markers")) if want_markers else None,
                                 C.byref(sdf.desc) if sdf is not None else None,
                                 _lib.ptr(R0), _lib.ptr(T0),
                                 _lib.ptr(out.get("pene_count")) if sdf is not None else None,
                                 _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "egx_lbs_forward")
        return out


    def joints55(self, xb: torch.Tensor, betas: torch.Tensor, frames_per_agent: int) -> torch.Tensor:
        """The 55 kinematic-tree joints [B,55,3] without the vertex pass (egx_lbs_joints)."""
        lib = _lib.load()
        B = int(xb.shape[0])
        A = int(betas.shape[0])
        if xb.dim() != 2 or xb.shape[1] != 93 or B == 0:
            raise ValueError(f"xb must be a non-empty [B,93], got {tuple(xb.shape)}")
        if betas.dim() != 2 or betas.shape[1] != 10 or A * frames_per_agent != B:
            raise ValueError(f"betas must be [B/frames_per_agent,10]; got {tuple(betas.shape)} for B={B}, fpa={frames_per_agent}")
        xb = xb.to(dtype=torch.float32).contiguous()
        betas = betas.to(dtype=torch.float32).contiguous()
        out = torch.empty(B, 55, 3, dtype=torch.float32, device=xb.device)
        ws = self.workspace(B)
        _lib.check(lib.egx_lbs_joints(self.handle, _lib.ptr(xb), _lib.ptr(betas), B, int(frames_per_agent), _lib.ptr(out),
                                      _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr()), "egx_lbs_joints")
        return out


class SMPLXOutput:
    """What `bm(return_verts=True, ...)` returns in the reference (fields used on this path)."""

    def __init__(self, vertices, joints):
        self.vertices = vertices
        self.joints = joints


class SMPLXParser:
    """models/baseops.py:271-598.  config = {'n_batch', 'device', 'marker_placement'} plus
    'body_models': {'male': BodyModelHandle, 'female': BodyModelHandle} (the reference builds them from
    the licensed npz files inside __init__; here the caller supplies synthetic or real model arrays)."""

    def __init__(self, config):
        for key, val in config.items():
            setattr(self, key, val)
        self.models: Dict[str, BodyModelHandle] = config["body_models"]
        any_model = next(iter(self.models.values()))
        self.marker = list(any_model.marker_vids)  # vertex ids, as in baseops.py:333-335

    def _bm(self, gender):
        if gender not in self.models:
            raise KeyError(f"no body model for gender {gender!r}")
        return self.models[gender]

    @staticmethod
    def _prep(betas, xb):
        if isinstance(xb, np.ndarray):
            xb = torch.from_numpy(np.ascontiguousarray(xb, dtype=np.float32)).cuda()
        if isinstance(betas, np.ndarray):
            betas = torch.from_numpy(np.ascontiguousarray(betas, dtype=np.float32))
        betas = betas.to(xb.device, torch.float32)
        betas = betas.reshape(-1, 10)
        return betas, xb.to(torch.float32)

    def forward_smplx(self, betas, gender, xb, to_numpy=True, output_type="markers"):
        """baseops.py:338-398.  output_type in markers | joints | all_joints | vertices | raw."""
        betas, xb = self._prep(betas, xb)
        B = xb.shape[0]
        fpa = B // betas.shape[0]
        bm = self._bm(gender)
        want_v = output_type in ("vertices", "raw")
        want_m = output_type == "markers"
        want_j = output_type in ("joints", "all_joints", "raw")
        if output_type not in ("markers", "joints", "all_joints", "vertices", "raw"):
            raise NotImplementedError("other output types are not supported")
        o = bm.forward(xb, betas, fpa, want_verts=want_v, want_joints=want_j, want_markers=want_m)
        if output_type == "raw":
            return SMPLXOutput(o["vertices"], o["joints"])
        res = {"markers": o.get("markers"), "joints": o["joints"][:, :22] if want_j else None,
               "all_joints": o.get("joints"), "vertices": o.get("vertices")}[output_type]
        return res.detach().cpu().numpy() if to_numpy else res

    def get_jts(self, betas, gender, xb, to_numpy=True):
        return self.forward_smplx(betas, gender, xb, to_numpy, "joints")

    def get_all_jts(self, betas, gender, xb, to_numpy=True):
        return self.forward_smplx(betas, gender, xb, to_numpy, "all_joints")

    def get_markers(self, betas, gender, xb, to_numpy=True):
        return self.forward_smplx(betas, gender, xb, to_numpy, "markers")

    def get_new_coordinate(self, betas, gender, xb, to_numpy=True):
        """baseops.py:465-490: canonical frame of every body of the batch -> (new_rotmat [b,3,3], new_transl [b,1,3]).
        The reference evaluates the whole body model for joints 0..2; here only the kinematic chain runs
        (egx_lbs_joints) followed by egx_canonical_frame (CanonicalCoordinateExtractor, baseops.py:214-225)."""
        lib = _lib.load()
        betas, xb = self._prep(betas, xb)
        B = xb.shape[0]
        j = self._bm(gender).joints55(xb, betas, B // betas.shape[0])
        R = torch.empty(B, 3, 3, dtype=torch.float32, device=xb.device)
        T = torch.empty(B, 1, 3, dtype=torch.float32, device=xb.device)
        _lib.check(lib.egx_canonical_frame(_lib.ptr(j), 55, B, _lib.ptr(R), _lib.ptr(T), _lib.current_stream_ptr()),
                   "egx_canonical_frame")
        return (R.cpu().numpy(), T.cpu().numpy()) if to_numpy else (R, T)

    def calc_calibrate_offset(self, bm, betas, body_pose, to_numpy=True):
        """baseops.py:494-534: delta_T [b,3] = pelvis of the body at zero global_orient / transl.  `bm` is a
        BodyModelHandle (the reference passes its smplx module)."""
        if isinstance(body_pose, np.ndarray):
            body_pose = torch.from_numpy(np.ascontiguousarray(body_pose, dtype=np.float32)).cuda()
        b = body_pose.shape[0]
        xb = torch.zeros(b, 93, dtype=torch.float32, device=body_pose.device)
        xb[:, 6:69] = body_pose.to(torch.float32)
        betas, xb = self._prep(betas, xb)
        d = bm.joints55(xb, betas, b // betas.shape[0])[:, 0].contiguous()
        return d.cpu().numpy() if to_numpy else d

    def update_transl_glorot(self, transf_rotmat, transf_transl, betas, gender, xb, to_numpy=True, inplace=True):
        """baseops.py:537-598: body parameters xb [b,93] re-expressed in the frame (transf_rotmat [b,3,3], transf_transl
        [b,1,3]).  Both of the reference's branches (scipy Rotation for numpy input, torchgeometry for tensors) compute
        glorot' = log(R^T exp(glorot)); here both go through egx_update_transl_glorot (the torchgeometry formulas)."""
        lib = _lib.load()
        src = xb
        betas_t, xbt = self._prep(betas, xb)
        xbt = xbt.contiguous()
        b = xbt.shape[0]
        dev = xbt.device
        as_t = lambda a: (torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) if isinstance(a, np.ndarray) else a).to(dev, torch.float32)
        R = as_t(transf_rotmat).reshape(-1, 3, 3).contiguous()
        T = as_t(transf_transl).reshape(-1, 3).contiguous()
        if R.shape[0] not in (1, b) or T.shape[0] != R.shape[0]:
            raise ValueError(f"transf_rotmat / transf_transl must hold 1 or {b} frames, got {R.shape[0]} / {T.shape[0]}")
        delta = self.calc_calibrate_offset(self._bm(gender), betas_t, xbt[:, 6:69], to_numpy=False)
        same = torch.is_tensor(src) and src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()
        out = xbt if (inplace and same) else torch.empty_like(xbt)
        _lib.check(lib.egx_update_transl_glorot(_lib.ptr(R), _lib.ptr(T), int(R.shape[0]), _lib.ptr(delta), _lib.ptr(xbt), b,
                                                _lib.ptr(out), _lib.current_stream_ptr()), "egx_update_transl_glorot")
        if to_numpy:
            res = out.cpu().numpy()
            if inplace and isinstance(src, np.ndarray):   # xb[:, :3] = transl; xb[:, 3:6] = glorot (baseops.py:583-585)
                src[:, :6] = res[:, :6]
                return src
            return res
        if inplace and torch.is_tensor(src) and not same:
            src[:, :6] = out[:, :6].to(src.device, src.dtype)
            return src
        return out
