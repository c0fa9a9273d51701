#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) for one kernel."""
import csv, glob, json, sys, os
root, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_render")
tot = {}
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if kernel not in row["Kernel_Name"]:
            continue
        k = row["Counter_Name"]
        tot.setdefault(k, []).append(float(row["Counter_Value"]))
out = {k: sum(v) / max(1, len(v)) if k in ("GRBM_GUI_ACTIVE",) else sum(v) for k, v in tot.items()}
n = {k: len(v) for k, v in tot.items()}
print(json.dumps({"per_counter_sum_over_dispatches": out, "rows": n}, indent=1))
