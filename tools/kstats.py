"""Prints the head of a rocprofv3 kernel_stats.csv: python tools/kstats.py <dir> [rows]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = sum(int(r["Calls"]) for r in rows)
print(f"{f}: kernel time {tot / 1e6:.1f} ms in {n} launches")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f'{r["Name"][:80]:80s} {int(r["Calls"]):8d} {float(r["TotalDurationNs"]) / 1e6:9.1f} ms {float(r["AverageNs"]) / 1e3:8.1f} us {float(r["Percentage"]):5.1f} %')
