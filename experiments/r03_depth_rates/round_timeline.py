#!/usr/bin/env python3
"""Per-dispatch timeline of the LAST kernel-9 launch in a rocprofv3 --kernel-trace CSV, beside the live-path counts the
library printed (RT_WF_PRINT_ALIVE=1): tools/round_timeline.py TRACE.csv LOG"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_wf_resolve' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
alive = [int(m.group(2)) for m in re.finditer(r"round (\d+) alive (\d+)", open(sys.argv[2]).read())]
# the log holds every launch; the last launch's rounds are the tail that starts at its "round 0"
starts = [i for i, m in enumerate(re.finditer(r"round (\d+) alive (\d+)", open(sys.argv[2]).read())) if m.group(1) == "0"]
alive = alive[starts[-1]:]
t0 = int(rows[a]['Start_Timestamp']); k = 0
for r in rows[a:b]:
    n = r['Kernel_Name']
    n = 'trace' if 'k_wf_trace' in n else 'shade' if 'k_wf_shade' in n else 'finish' if 'finish' in n else 'resolve' if 'resolve' in n else None
    if not n: continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    extra = ''
    if n == 'trace' and k < len(alive):
        extra = f" rays {alive[k]} = {alive[k] / dur / 1e6:.2f} Grays/s"; k += 1
    print(f"{n:8s} start {(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} dur {dur:7.3f} ms{extra}")
