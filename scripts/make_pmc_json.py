#!/usr/bin/env python3
"""profiles/<round>_lbs_pmc_mode<m>.json from the counters.txt scripts/pmc_lbs.sh wrote.
usage: make_pmc_json.py <counters.txt> <round> <blend mode> <bases MB>"""
import json, os, sys
src, rnd, mode, bases_mb = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
per = {}
kernel = None
for ln in open(src):
    p = ln.split()
    if len(p) >= 4 and "lbs_fused" in ln:
        kernel = " ".join(p[:-3]).replace("void ", "")
        per[p[-3]] = float(p[-1])
alg = bases_mb * 1e6 + 10240 * (412 + 2332)
d = {"kernel": f"{kernel} (blend mode {mode})",
     "command": "bash scripts/pmc_lbs.sh (rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python scripts/prof_lbs.py sdf)",
     "config": {"agents": 512, "bodies_per_launch": 10240, "num_verts": 10475, "sdf": "single_box 256^3"},
     "per_launch": per, "derived": {}}
dv = d["derived"]
if "FETCH_SIZE" in per:
    dv["hbm_side_bytes_per_launch"] = int(per["FETCH_SIZE"] * 1024 * 2 + per.get("WRITE_SIZE", 0) * 1024)
    dv["note"] = ("FETCH_SIZE (KiB) x 2 (gfx950 correction for 16-byte-per-lane streaming reads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE; "
                  "both count L2-miss-side (fabric / Infinity Cache) requests, not only HBM: the whole operand set fits the 256 MiB Infinity Cache")
if "TCC_HIT_sum" in per:
    dv["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "SQ_BUSY_CU_CYCLES" in per:
    dv["mfma_pipe_utilisation"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * per["SQ_BUSY_CU_CYCLES"]) if per["SQ_BUSY_CU_CYCLES"] else None
if "TCP_TCC_READ_REQ_sum" in per:
    dv["l2_to_l1_bytes_per_launch"] = int(per["TCP_TCC_READ_REQ_sum"] * 128)
    dv["l2_to_l1_note"] = "TCP_TCC_READ_REQ_sum x 128 B (request size calibrated on the kernel's own operand traffic); scripts/ubench/l2_bw.hip sustains 32 TB/s from L2"
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in per:
        dv["l1_hit_rate"] = 1.0 - per["TCP_TCC_READ_REQ_sum"] / per["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "SQ_INSTS_VALU" in per:
    dv["valu_wave_instructions_per_simd"] = per["SQ_INSTS_VALU"] / 1024.0
dv["algorithmic_bytes_per_launch"] = int(alg)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{rnd}_lbs_pmc_mode{mode}.json")
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(dv, indent=1))
