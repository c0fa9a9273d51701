#!/usr/bin/env python3
"""Per-kernel counter sums of ONE rt_render launch from the PMC passes of tools/profile_pmc.sh, written to
profiles/roofline_latest.json, which bench.py combines with the kernel durations it measures itself (HIP events).

Counted per kernel (all dispatches of the launch summed): VALU / SALU / VMEM wave-instructions, VALU lane
utilisation (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64), wave cycles and wait cycles, L2 hits / misses,
L1 accesses, and the memory-side byte counters.  About those (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE
are the L2's fabric-side requests, so they INCLUDE reads served by the Infinity Cache — for a scene that lives in it
they are L2-miss traffic, not HBM traffic; FETCH_SIZE counts 128-byte requests as 64, so it is exact for 64-byte
requests and half the truth for wide streaming reads: both readings are written (raw, x2), the width of this path's
16-byte divergent gathers being uncalibrated.
Usage: roofline_from_pmc.py OUT_JSON RNG:PMC_DIR:KERNEL[,KERNEL...] ...   (LAUNCHES per PMC run is 1)"""
import csv, glob, json, os, subprocess, sys

out_path, specs = sys.argv[1], sys.argv[2:]
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    commit = None
doc = {"commit": commit, "kernels": {}, "note": __doc__.split("Usage:")[0].strip()}
for spec in specs:
    rng, root, kernels = spec.split(":")
    rows = []
    for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
        rows += list(csv.DictReader(open(f)))
    for kernel in kernels.split(","):
        tot, disp = {}, set()
        for row in rows:
            if kernel in row["Kernel_Name"]:
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if row["Counter_Name"] == "SQ_INSTS_VALU":
                    disp.add(row["Dispatch_Id"])
        if not tot:
            continue
        g = lambda k: tot.get(k, 0.0)
        doc["kernels"][f"{rng}:{kernel}"] = {
            "rng": rng, "kernel": kernel, "pmc_dir": root, "dispatches": len(disp),
            "valu_wave_insts": g("SQ_INSTS_VALU"), "salu_wave_insts": g("SQ_INSTS_SALU"), "vmem_wave_insts": g("SQ_INSTS_VMEM"),
            "lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") / 64 if g("SQ_ACTIVE_INST_VALU") else None,
            "wait_any_over_wave_cycles": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
            "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else None,
            "l1_accesses": g("TCP_TOTAL_CACHE_ACCESSES_sum"), "l2_read_requests": g("TCP_TCC_READ_REQ_sum"),
            "fetch_bytes_raw": g("FETCH_SIZE") * 1024.0, "write_bytes": g("WRITE_SIZE") * 1024.0,
        }
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(doc["kernels"], indent=1))
