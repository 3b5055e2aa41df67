#!/usr/bin/env python3
"""Development aid: per-kernel durations and the idle gap before each kernel from a rocprofv3 --kernel-trace CSV.
    python scripts/trace_gaps.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name] += e - s
    if prev_end is not None:
        gap[name] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
    cnt[name] += 1
tot_d, tot_g = sum(dur.values()), sum(gap.values())
print(f"{len(rows)} kernels: busy {tot_d/1e3:.1f} us, gaps {tot_g/1e3:.1f} us, span {(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3:.1f} us")
for name in sorted(cnt, key=lambda n: -(dur[n] + gap[n])):
    print(f"{cnt[name]:6d} x {name:60s} dur {dur[name]/cnt[name]/1e3:8.2f} us   gap before {gap[name]/cnt[name]/1e3:7.2f} us")
