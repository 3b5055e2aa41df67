#!/bin/bash
# Regenerate egogen_amd/data/tunableop_gfx950.csv: PyTorch TunableOp picks, per GEMM shape of the PPO update, the fastest of the
# hipBLASLt / rocBLAS solutions on THIS GPU and library build (run on the GPU box; the table is only honoured when its
# validator lines - torch, HIP, hipBLASLt, rocBLAS versions and the GCN arch - match the running stack).
#   bash scripts/tune_gemms.sh            (local minibatch rows 256, 128, 64, 32 = 1, 2, 4, 8 ranks of the strong-scaling bench)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/tune
mkdir -p "$OUT"
export EGX_TUNED_GEMM=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$OUT/tunableop.csv
for bs in 256 128 64 32; do
  timeout 900 python $R/bench.py --no-cpu-baseline --extra-configs 0 --steps 2 --warmup 1 --batch-size $bs < /dev/null > "$OUT/bench_$bs.json" 2> "$OUT/bench_$bs.err"
  echo "batch $bs rc=$? lines=$(wc -l < $OUT/tunableop0.csv 2>/dev/null)"
done
cp "$OUT/tunableop0.csv" "$OUT/tunableop_gfx950.csv" && cat "$OUT/tunableop_gfx950.csv"
