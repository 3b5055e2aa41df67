#!/bin/bash
# PMC passes over scripts/prof_lbs.py (one counter set per pass, as MI355X_MICROARCH.md prescribes); run on the GPU box:
#   bash scripts/pmc_lbs.sh <outdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_lbs}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
SETS=${PMC_SETS:-"FETCH_SIZE WRITE_SIZE TCC_HIT_sum,TCC_MISS_sum GRBM_GUI_ACTIVE,SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD,SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES,SQ_WAIT_INST_ANY"}
for cs in $SETS; do
  set=${cs//,/ }
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $R/scripts/prof_lbs.py sdf > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" >> "$OUT/counters.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "lbs_fused" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, len(v), sum(v) / len(v))
PY
  else
    echo "pass $i ($set): no counter file (rc or timeout)" >> "$OUT/counters.txt"; tail -3 /tmp/pmc_$i.log >> "$OUT/counters.txt"
  fi
done
cat "$OUT/counters.txt"
