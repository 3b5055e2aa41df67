"""Development aid: per-kernel averages of a rocprofv3 `--pmc` counter_collection CSV for the dense / GRU / packing kernels.
    python scripts/pmc_by_kernel.py <counter_collection.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:44]
    acc[(k, r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    if "dense3" in k or "gru3" in k or "pack3_table" in k:
        print(f"{k:46s} grid {g:>8s} {c:22s} n={len(v):3d} avg {sum(v)/len(v):.4g}")
