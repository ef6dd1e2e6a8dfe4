#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (per launch)."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("df_"): continue
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    print(k)
    for c in sorted(rows[k]):
        v = rows[k][c]
        print("    %-24s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
