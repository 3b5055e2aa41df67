"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/egogen_hip.h declares (no compute calls - there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "egogen_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(egx_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_bound_and_exported():
    from egogen_amd import _lib
    declared = _declared_symbols()
    assert declared, "no symbols parsed from the header"
    assert set(declared) == set(_lib.SIGNATURES.keys()), set(declared) ^ set(_lib.SIGNATURES.keys())
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()  # raises if any declared symbol is missing from the .so
    assert lib.egx_version() >= 1


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under egogen_amd/ or crowd_ppo/ may reference it."""
    bad = []
    for base in ("egogen_amd", "crowd_ppo"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "oracle/" in src:
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_training_nodes_call_no_library_gemm():
    """The autograd nodes and training operators run their matrix products through egx_gemm3 / the update chain: no
    torch GEMM entry point (hipBLASLt / rocBLAS behind it) is spelled in those sources."""
    pat = re.compile(r"torch\.(addmm|mm|bmm|baddbmm|matmul|einsum|tensordot)\(|\.addmm_\(|F\.linear\(")
    bad = []
    for fn in ("fused_ops.py", "ppo_policy.py", "train_predictor.py", "trainer.py"):
        src = open(os.path.join(ROOT, "egogen_amd", fn)).read()
        bad += [(fn, m.group(0)) for m in pat.finditer(src)]
    src = open(os.path.join(ROOT, "egogen_amd", "train_regressor.py")).read()
    bad += [("train_regressor.py", m.group(0)) for m in pat.finditer(src)]
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from egogen_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EgxError):
        _lib.load()


def test_struct_layouts_match_the_header(tmp_path):
    """Every struct of include/egogen_hip.h has the same size and field offsets in the ctypes mirror (compiled with the
    host C compiler from the header itself, so the binding cannot drift from the ABI unnoticed)."""
    import ctypes as C
    import subprocess
    from egogen_amd import _lib
    pairs = {"egx_body_model_host": _lib.BodyModelHost, "egx_sdf_grid": _lib.SdfGrid, "egx_linear_desc": _lib.LinearDesc,
             "egx_prior_weights": _lib.PriorWeights, "egx_policy_weights": _lib.PolicyWeights,
             "egx_vposer_weights": _lib.VposerWeights, "egx_env_config": _lib.EnvConfig, "egx_env_scenes": _lib.EnvScenes,
             "egx_env_state": _lib.EnvState, "egx_env_step_io": _lib.EnvStepIO, "egx_env_reset_io": _lib.EnvResetIO}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "egogen_hip.h"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in pairs.items():
        assert int(got[cname]) == C.sizeof(ct), (cname, got[cname], C.sizeof(ct))
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)


def test_prototype_arities_match_the_binding():
    """Number of parameters of every prototype in the header == number of ctypes argtypes of its binding."""
    from egogen_amd import _lib
    txt = open(os.path.join(ROOT, "include", "egogen_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = dict(re.findall(r"\b(egx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S))
    assert set(protos) == set(_lib.SIGNATURES)
    for name, params in protos.items():
        params = params.strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(_lib.SIGNATURES[name][1]), (name, n, len(_lib.SIGNATURES[name][1]))
