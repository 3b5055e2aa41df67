# Development aid (GPU box): fused LBS kernel of several library builds, interleaved
#   bash scripts/ab_lbs_variants.sh "base hoist nobar both" [rounds] [test-lib]
set -u
cp egogen_amd/libegogen_hip.so /tmp/lib_product.so
ROUNDS=${2:-3}
for r in $(seq 1 $ROUNDS); do for v in $1; do
  cp ab_libs/lib_$v.so egogen_amd/libegogen_hip.so
  echo "round $r $v: $(EGX_BENCH_MODES=3 EGX_BENCH_AGENTS=${AGENTS:-512} timeout 300 python scripts/bench_lbs.py 2>&1 | grep -E 'picks\+sdf' | grep -v verts | sed "s/blend.*fixups/fixups/" | tr '\n' ';')"
done; done
if [ -n "${3:-}" ]; then
  cp ab_libs/lib_$3.so egogen_amd/libegogen_hip.so
  timeout 1500 python -m pytest tests/test_lbs_gpu.py tests/test_env_reference_goldens.py -m gpu -x -q 2>&1 | tail -5
fi
cp /tmp/lib_product.so egogen_amd/libegogen_hip.so
