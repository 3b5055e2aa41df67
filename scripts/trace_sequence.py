"""Development aid: the kernel sequence (durations, idle gap before each) between the last two launches of a marker kernel.
    python scripts/trace_sequence.py <kernel_trace.csv> [marker substring, default gather_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "gather_rows"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
prev = None
tot = gaps = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = max(prev or 0, e)
    tot += e - s
    gaps += max(gap, 0)
    print(f"{r['Kernel_Name'].split('(')[0][:60]:60s} dur {(e-s)/1e3:8.2f} us  gap {gap:7.2f}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
print("kernels", hi - lo, "sum of durations", tot / 1e3, "us; gaps", gaps / 1e3, "us; span", (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3)
