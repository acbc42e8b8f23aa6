"""Largest kernel durations and inter-kernel gaps in a rocprofv3 kernel trace CSV."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dur = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:40]) for r in rows)
gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"][:30], b["Kernel_Name"][:30]) for a, b in zip(rows, rows[1:]))
print("launches", len(rows), "longest kernels (us):", [(round(d), n) for d, n in dur[-3:]])
print("largest gaps (us):", [(round(g), a, b) for g, a, b in gaps[-4:]])
