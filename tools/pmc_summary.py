"""Summarise a rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv in one directory): per kernel, mean
counter values and mean duration per launch, plus two derived figures the guide defines:
  clock_GHz = GRBM_GUI_ACTIVE / 8 / duration   (sum over the 8 XCDs; trust it on launches >= 0.3 ms)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
Usage: python tools/pmc_summary.py <dir> [name-substring]"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in cc:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if want not in name:
            continue
        key = name.split("(")[0][:60]
        acc[key][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
out = {}
for k, cs in acc.items():
    row = {}
    ids = None
    for c, vals in cs.items():
        row[c] = sum(v for _, v in vals) / len(vals)
        ids = [i for i, _ in vals]
    ds = [dur[i] for i in ids if i in dur]
    row["launches"] = len(ids)
    if ds:
        row["us_per_launch"] = sum(ds) / len(ds)
        if "GRBM_GUI_ACTIVE" in row:
            row["clock_GHz"] = row["GRBM_GUI_ACTIVE"] / 8 / (row["us_per_launch"] * 1e3)
    if "GRBM_GUI_ACTIVE" in row and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
        row["mfma_busy"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_LDS_BANK_CONFLICT" in row and row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict_frac"] = row["SQ_LDS_BANK_CONFLICT"] / row["SQ_LDS_IDX_ACTIVE"]
    out[k] = row
print(json.dumps(out, indent=1))
