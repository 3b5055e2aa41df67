#!/bin/bash
# Development aid (GPU box): time the fused LBS kernel with several builds of the library kept under build/.
#   bash scripts/ab_lbs.sh build/lib_a.so build/lib_b.so ...
set -u
cp egogen_amd/libegogen_hip.so /tmp/lib_product.so
for f in "$@"; do
  echo "== $f"
  cp "$f" egogen_amd/libegogen_hip.so
  EGX_BENCH_MODES=${EGX_BENCH_MODES:-2} timeout 300 python scripts/bench_lbs.py 2>&1 | grep -E "A=512 .*(picks|verts)" | grep -v "verts"
done
cp /tmp/lib_product.so egogen_amd/libegogen_hip.so
