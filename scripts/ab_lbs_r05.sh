# Development aid (GPU box): fused LBS kernel, mode 3 vs 2, with the epilogue / the GEMM skipped (EGX_LBS_DBG, development build)
set -u
cp egogen_amd/libegogen_hip.so /tmp/lib_product.so
cp ${1:-ab_libs/lib_b_batch16.so} egogen_amd/libegogen_hip.so
for mode in 3 2; do for dbg in 0 1 2; do
  echo "mode $mode dbg $dbg (1 = no epilogue, 2 = no GEMM): $(EGX_LBS_DBG=$dbg EGX_BENCH_MODES=$mode timeout 300 python scripts/bench_lbs.py 2>&1 | grep -E 'A=512 .*picks\+sdf' | sed 's/blend.*//')"
done; done
cp /tmp/lib_product.so egogen_amd/libegogen_hip.so
