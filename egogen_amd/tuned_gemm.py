"""Tuned solution table for the library GEMMs of the PPO update (`torch.addmm / mm` inside fused_ops.LinearFn / ResMLPFn /
GRUSeqFn): PyTorch's TunableOp, read-only.  `egogen_amd/data/tunableop_gfx950.csv` was produced on an MI355X by
`scripts/tune_gemms.sh` (every GEMM shape of the update at 256 / 128 / 64 / 32 local minibatch rows timed over the hipBLASLt
and rocBLAS solutions); TunableOp honours it only when its validator lines (torch, HIP, hipBLASLt, rocBLAS versions, GCN arch)
match the running stack and otherwise keeps the libraries' own heuristics.  No tuning happens at run time.
`EGX_TUNED_GEMM=0` disables it."""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tunableop_gfx950.csv")
_state = {"done": False, "active": False}


def enable() -> bool:
    """Idempotent; returns whether the table is in use."""
    if _state["done"]:
        return _state["active"]
    _state["done"] = True
    if os.environ.get("EGX_TUNED_GEMM", "1") != "1" or os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        return False       # switched off, or the user drives TunableOp through its own environment variables
    if not (torch.cuda.is_available() and os.path.exists(TABLE)):
        return False
    try:
        t = torch.cuda.tunable
        t.enable(True)
        t.tuning_enable(False)
        _state["active"] = bool(t.read_file(TABLE))
        if not _state["active"]:
            t.enable(False)
    except Exception:      # an older torch without the API: the libraries' own choices
        _state["active"] = False
    try:
        torch.cuda.tunable.write_file_on_exit(False)   # nothing is tuned here, so there is nothing to write back
    except Exception:
        pass
    return _state["active"]
