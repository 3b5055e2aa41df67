#!/usr/bin/env python3
"""Markdown table of a rocprofv3 kernel_stats.csv per benchmark cycle.  usage: kernel_table.py <kernel_stats.csv> <cycles> [rows]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ncyc = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 22
tot = sum(float(r["TotalDurationNs"]) for r in rows)
cat, calls = collections.Counter(), collections.Counter()
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "")
    if n.startswith("Cijk"): k = "library GEMMs (Cijk_*)"
    elif "egx_" in n: k = n.split("(")[0].replace("void ", "")[:48]
    elif "at::native" in n: k = "torch " + n.split("at::native::")[1].split("<")[0][:40]
    else: k = n[:48]
    cat[k] += float(r["TotalDurationNs"]); calls[k] += int(r["Calls"])
print("| share | ms / cycle | launches / cycle | avg us | kernel |\n|---|---|---|---|---|")
for k, v in cat.most_common(top):
    print(f"| {v/tot*100:.2f} % | {v/ncyc/1e6:.3f} | {calls[k]/ncyc:.1f} | {v/calls[k]/1e3:.1f} | `{k}` |")
egx = sum(v for k, v in cat.items() if "egx_" in k)
print(f"\nhand-written kernels (`egx_*`): {egx/tot*100:.1f} % of GPU time; summed kernel time {tot/ncyc/1e6:.2f} ms per cycle; {sum(calls.values())/ncyc:.0f} launches per cycle")
