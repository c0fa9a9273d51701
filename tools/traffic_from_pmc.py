#!/usr/bin/env python3
"""HBM traffic per rt_render launch from the tcc1/tcc2 PMC passes (tools/profile_pmc.sh), written to
profiles/traffic_latest.json for bench.py's roofline.traffic.

rocprofv3's FETCH_SIZE / WRITE_SIZE are KiB summed over the dispatch.  MI355X_MICROARCH.md (HBM):
on gfx950 FETCH_SIZE reports half the bytes of wide reads (128-B requests tallied at 64 B), so it is
doubled; WRITE_SIZE is exact.  The guide calls other access widths (our 16-B-per-lane divergent
gathers) uncalibrated, so the doubled figure is an upper bound on the HBM/Infinity-Fabric read bytes.
One rt_render launch is one dispatch of the megakernel (kernel 8) or every k_wf_* dispatch of the
wavefront pipeline (kernel 9): the counters of all dispatches whose name contains KERNEL_SUBSTRING
are summed and divided by LAUNCHES.
Usage: traffic_from_pmc.py PMC_DIR KERNEL_SUBSTRING OUT_JSON RNG LAUNCHES
"""
import csv, glob, json, os, subprocess, sys
root, kernel, out, rng, launches = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
disp = {"FETCH_SIZE": set(), "WRITE_SIZE": set()}
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"] and row["Counter_Name"] in tot:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Counter_Name"]].add(row["Dispatch_Id"])
fetch = tot["FETCH_SIZE"] / launches * 1024.0
write = tot["WRITE_SIZE"] / launches * 1024.0
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    commit = None
doc = {"kernel": kernel, "rng": rng, "fetch_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch, "write_bytes": write,
       "hbm_bytes_per_launch": 2 * fetch + write, "launches_averaged": launches,
       "dispatches_per_launch": len(disp["FETCH_SIZE"]) / launches, "pmc_dir": root, "commit": commit,
       "note": "FETCH_SIZE KiB x1024 x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x1024; "
               "upper bound for 16-B divergent gathers (uncalibrated width); summed over every dispatch of one rt_render launch"}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc))
