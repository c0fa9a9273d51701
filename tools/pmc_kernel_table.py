#!/usr/bin/env python3
"""Per-kernel PMC table from tools/profile_pmc.sh output: sums counters per kernel name."""
import csv, glob, os, sys, collections
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add((f, row["Dispatch_Id"]))
for k, d in tot.items():
    if not any(s in k for s in ("k_wf", "k_render", "k_trace")):
        continue
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    def r(a, b): return d.get(a, 0) / d[b] if d.get(b) else float("nan")
    print(k)
    print("  VALU lane util %.3f  VALUBusy %.1f%%  cycles/xcd %.3g" % (r("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU") / 64, 100 * d.get("SQ_ACTIVE_INST_VALU", 0) / 256 / g if g else 0, g))
    print("  insts valu %.3g salu %.3g vmem %.3g lds %.3g" % (d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_VMEM", 0), d.get("SQ_INSTS_LDS", 0)))
    print("  wait_any %.2f wait_inst %.2f active %.2f | TA %.2f TD %.2f L2hit %.2f" % (r("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), r("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), r("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"), d.get("TA_TA_BUSY_sum", 0) / 256 / g if g else 0, d.get("TD_TD_BUSY_sum", 0) / 256 / g if g else 0, d.get("TCC_HIT_sum", 0) / max(1, d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0))))
