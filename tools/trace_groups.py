import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
g = {}
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))
    g.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in g.items():
    if "wgrad" in k[0]:
        print(f"{k[0]:42s} grid {k[1]:>8}  n {len(v):4d}  median {statistics.median(v):8.1f} us")
