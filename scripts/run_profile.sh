#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (run on the GPU box): bash scripts/run_profile.sh <tag>
# EGX_PROFILE_FLAGS adds bench.py flags (e.g. "--scene box" = BASELINE configs[2] verbatim)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $R/bench.py --no-cpu-baseline --extra-configs 0 --steps 10 ${EGX_PROFILE_FLAGS:-} > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.log" < /dev/null
echo "rocprofv3 rc=$?"
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then
  cp "$f" "$OUT/kernel_stats.csv"
  head -45 "$OUT/kernel_stats.csv" | cut -c1-180
else
  echo "no kernel_stats.csv found"; find /tmp/prof_$TAG -type f 2>/dev/null | head; tail -5 "$OUT/bench_under_rocprof.log"
fi
