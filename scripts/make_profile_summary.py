#!/usr/bin/env python3
"""profiles/<round>_final_summary.md from the committed rocprofv3 kernel stats + bench JSON lines.
usage: make_profile_summary.py r02 <cycles under rocprofv3>"""
import csv, collections, json, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r01"
rows = list(csv.DictReader(open(os.path.join(R, f"{RND}_final_bench_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
cat, calls = collections.Counter(), collections.Counter()
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "")
    if n.startswith("Cijk"): k = "library GEMMs (Cijk_*: hipBLASLt / rocBLAS Tensile kernels)"
    elif "egx_" in n: k = n.split("(")[0].replace("void ", "")[:48]
    elif "at::native" in n: k = "torch " + n.split("at::native::")[1].split("<")[0][:40]
    else: k = n[:48]
    cat[k] += float(r["TotalDurationNs"]); calls[k] += int(r["Calls"])
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = json.load(open(os.path.join(R, f"{RND}_final_bench.json")))
u = json.loads(open(os.path.join(R, f"{RND}_final_bench_under_rocprof.json")).read())
lbs = max((r for r in rows if "egx_lbs_fused3" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
L = [f"# Round {int(RND[1:])} final profile (1x MI355X)", "",
     f"Command: `bash scripts/run_profile.sh` = `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --extra-configs 0 --steps 10`",
     f"(per-cycle figures below divide by {ncyc}).  Un-profiled run of `python bench.py`: `{RND}_final_bench.json`.", "",
     f"* un-profiled: **{b['value']:.0f} env-steps/s**, {b['ms_per_step']:.2f} ms per 2048-transition cycle; fused LBS kernel {b['roofline']['avg_launch_ms']:.3f} ms per launch (HIP events) = {b['roofline']['achieved']:.1f} TFLOP/s fp32-equivalent, frac {b['roofline']['frac']:.3f} of {b['roofline']['peak']:.1f} ({b['roofline'].get('peak_note', '16-bit MFMA peak / products per fp32 product')}); traffic {b['roofline']['traffic']/1e9:.2f} GB per launch (PMC)",
     f"* under rocprofv3: {u['value']:.0f} env-steps/s; LBS kernel: rocprofv3 average {float(lbs['AverageNs'])/1e6:.3f} ms over {lbs['Calls']} launches, HIP-event average in the same run {u['roofline']['avg_launch_ms']:.3f} ms",
     f"* cpu_baseline: {b['cpu_baseline']['value']:.2f} env-steps/s on {b['cpu_baseline']['cores']} threads ({b['cpu_baseline']['sample']})", "",
     "| share | ms / cycle | launches / cycle | avg us | kernel |", "|---|---|---|---|---|"]
for k, v in cat.most_common(24):
    L.append(f"| {v/tot*100:.2f} % | {v/ncyc/1e6:.3f} | {calls[k]/ncyc:.1f} | {v/calls[k]/1e3:.1f} | `{k}` |")
# derived rates of kernels whose algorithmic work per launch is fixed by the benchmark configuration
def avg_us(sub):
    r = [x for x in rows if sub in x["Name"]]
    return sum(float(x["TotalDurationNs"]) for x in r) / max(1, sum(int(x["Calls"]) for x in r)) / 1e3
n_flat = 13_168_001 + 28 * 32          # parameters + alignment padding of the flat buffers (upper bound of the padding)
adam_bytes = 7 * 4 * n_flat            # read p, g, m, v; write p, m, v
reg_flop = 2.0 * 9216 * 3 * (370 * 128 + 20 * 128 * 128 + 128 * 159)
reg3 = any("egx_regressor3" in x["Name"] for x in rows)
reg_name = "egx_regressor3" if reg3 else "egx_regressor_fused"
reg_bound = ("bf16 MFMA, six partial products per fp32 product: 416.7 TFLOP/s fp32-equivalent ceiling / LDS + L2 weight stream" if reg3
             else "fp32 MFMA 157.3 / L2 weight stream (64 KB per layer and workgroup)")
extra = ["", "| kernel | work per launch | avg us | rate | bound |", "|---|---|---|---|---|",
         f"| `egx_adamw_flat_kernel` | {adam_bytes/1e6:.0f} MB (p, g, m, v in; p, m, v out) | {avg_us('egx_adamw_flat'):.1f} | {adam_bytes/avg_us('egx_adamw_flat')/1e6:.2f} TB/s | HBM (8 TB/s peak, ~6.3 achievable) |",
         f"| `{reg_name}_kernel` | {reg_flop/1e9:.1f} GFLOP fp32-equivalent (9216 rows x 66 layers) | {avg_us(reg_name):.1f} | {reg_flop/avg_us(reg_name)/1e6:.1f} TFLOP/s | {reg_bound} |"]
egx = sum(v for k, v in cat.items() if "egx_" in k)
L += extra
L += ["", f"hand-written kernels (`egx_*`): {egx/tot*100:.1f} % of GPU time; summed kernel time {tot/ncyc/1e6:.2f} ms per cycle (one stream: the sum is the GPU-busy part of the wall time).",
      f"PMC passes of the LBS kernel: `{RND}_lbs_pmc_mode*.json`; experiments of the round: " + ", ".join(f"`{f}`" for f in sorted(os.listdir(R)) if f.startswith(RND) and f.endswith(".md") and "final" not in f) + "; micro-benchmarks: `r01_ubench.md`."]
open(os.path.join(R, f"{RND}_final_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[5:9]))
