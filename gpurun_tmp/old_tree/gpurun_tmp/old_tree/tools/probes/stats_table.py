"""Print a rocprofv3 --stats kernel table per pass:  python tools/probes/stats_table.py <dir> <passes> [rows]"""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per pass: {tot / n / 1e6:.3f} ms in {sum(int(r['Calls']) for r in rows) // n} launches")
for r in rows[:top]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']) / n:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us = {float(r['TotalDurationNs']) / n / 1e3:8.1f} us/pass")
