"""Compact view of a rocprofv3 *_kernel_stats.csv: name[:70], calls, total ms, avg us, %."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = r["Name"].replace("void at::native::", "").replace("(anonymous namespace)::", "")
    print(f"{n[:78]:78s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
