"""Print the top kernels of a rocprofv3 *kernel_stats.csv.  Usage: python tools/kstats.py <dir-or-csv> <steps> [top]"""
import csv
import glob
import os
import sys

path, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {int(r['Calls']) / steps:7.1f} calls {float(r['Percentage']):5.1f}% {r['Name'][:120]}")
print(f"total {tot / steps / 1e6:.2f} ms/step  ({path})")
