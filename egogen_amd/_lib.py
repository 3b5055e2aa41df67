"""ctypes binding of libegogen_hip.so (the C ABI declared in include/egogen_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libegogen_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class EgxError(RuntimeError):
    pass


class BodyModelHost(C.Structure):
    _fields_ = [
        ("num_verts", C.c_int),
        ("v_template_host", C.c_void_p), ("shapedirs_host", C.c_void_p), ("posedirs_host", C.c_void_p),
        ("J_regressor_host", C.c_void_p), ("parents_host", C.c_void_p), ("lbs_weights_host", C.c_void_p),
        ("hand_comps_l_host", C.c_void_p), ("hand_comps_r_host", C.c_void_p),
        ("hand_mean_l_host", C.c_void_p), ("hand_mean_r_host", C.c_void_p),
        ("extra_vids_host", C.c_void_p), ("lmk_vids_host", C.c_void_p), ("lmk_bary_host", C.c_void_p),
        ("num_markers", C.c_int), ("marker_vids_host", C.c_void_p),
        ("num_feet", C.c_int), ("feet_vids_host", C.c_void_p),
    ]


class SdfGrid(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("d2", C.c_int),
                ("center", C.c_float * 3), ("scale", C.c_float)]


# name -> (restype, argtypes); must list every symbol of include/egogen_hip.h
SIGNATURES = {
    "egx_last_error": (C.c_char_p, []),
    "egx_version": (C.c_int, []),
    "egx_body_model_create": (C.c_int, [C.POINTER(BodyModelHost), C.POINTER(C.c_void_p)]),
    "egx_body_model_destroy": (None, [C.c_void_p]),
    "egx_body_model_num_verts": (C.c_int, [C.c_void_p]),
    "egx_body_model_nnz": (C.c_int, [C.c_void_p]),
    "egx_lbs_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "egx_lbs_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(SdfGrid), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "egx_sdf_sample": (C.c_int, [C.POINTER(SdfGrid), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Load the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EgxError(f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C egogen_amd/csrc`; there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().egx_last_error()
        raise EgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
