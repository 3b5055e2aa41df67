#!/bin/bash
# GPU box: K2 (standalone calc_sdf) timing, kernel trace and HBM-side traffic counters -> gpurun_out/sdf_k2/
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/sdf_k2; mkdir -p $O
timeout 200 python scripts/bench_sdf.py > $O/bench_sdf.jsonl 2> $O/bench_sdf.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sdfprof
EGX_SDF_ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sdfprof/trace -o sdf -- python $R/scripts/bench_sdf.py > $O/trace.log 2>&1
for c in "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $c | tr ' ' '_')
  EGX_SDF_ITERS=2 EGX_SDF_AGENTS=64 timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/sdfprof/pmc_$n -o sdf -- python $R/scripts/bench_sdf.py > $O/pmc_$n.log 2>&1
done
cd $R
find /tmp/sdfprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python - <<'PY'
import csv, glob, os
O = "gpurun_out/sdf_k2"
out = open(O + "/pmc_summary.txt", "w")
for f in glob.glob("/tmp/sdfprof/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = {}
    for r in csv.DictReader(open(f)):
        if "sdf_sample" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        line = f"{k} launches {len(v)} mean {sum(v) / len(v):.1f} last {v[-1]:.1f}"
        print(line); out.write(line + "\n")
PY
