#!/bin/bash
# Regenerate every fixture that is produced by EXECUTING the reference (scripts/gen_env_goldens.py: all six targets; needs
# /root/reference, i.e. the build container) into a scratch directory and compare array by array with tests/golden/.
#   bash scripts/regen_and_diff_goldens.sh            -> prints "<file>: N arrays, 0 differing" per fixture, exit 1 on any difference
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/egx_goldens.XXXXXX)
cd "$R"
EGX_GOLDEN_OUT=$T python scripts/gen_env_goldens.py ${1:-all} > "$T/gen.log" 2>&1 || { tail -20 "$T/gen.log"; exit 2; }
python - "$T" "$R/tests/golden" <<'PY'
import glob, os, sys
import numpy as np
new, old = sys.argv[1], sys.argv[2]
bad = 0
for f in sorted(glob.glob(os.path.join(new, "*.npz"))):
    a, b = np.load(f, allow_pickle=True), np.load(os.path.join(old, os.path.basename(f)), allow_pickle=True)
    keys = sorted(set(a.files) | set(b.files))
    diff = [k for k in keys if k not in a.files or k not in b.files or a[k].dtype != b[k].dtype or a[k].shape != b[k].shape
            or not np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f")]
    print(f"{os.path.basename(f)}: {len(keys)} arrays, {len(diff)} differing" + (f" {diff[:5]}" if diff else ""))
    bad += len(diff)
sys.exit(1 if bad else 0)
PY
