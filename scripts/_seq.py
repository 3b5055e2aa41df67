import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last minibatch: find the last gather_rows kernel
idx = [i for i, r in enumerate(rows) if "gather_rows" in r["Kernel_Name"]]
lo = idx[-2]; hi = idx[-1]
prev = None
tot = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    tot += (e - s)
    print(f"{r['Kernel_Name'].split('(')[0][:50]:50s} dur {(e-s)/1e3:7.2f} us  gap {gap:6.2f}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
print("sum of durations", tot / 1e3, "us; span", (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3)
