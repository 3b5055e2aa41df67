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


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from egogen_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EgxError):
        _lib.load()
