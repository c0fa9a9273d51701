#!/usr/bin/env python3
"""A/B of one bench flag: tools/ab_flag.py FLAG v1,v2[,...] REPEATS [other bench flags...]; one bench.py child per value and repeat."""
import json, os, subprocess, sys
flag, values, repeats = sys.argv[1], sys.argv[2].split(","), int(sys.argv[3])
rest = sys.argv[4:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(repeats):
    for v in values:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), flag, v] + rest, capture_output=True, text=True, timeout=400)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(out.returncode)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = (d.get("roofline") or {}).get("kernel_ms_by_kernel") or {}
        print(flag, v, "Mrays/s %.1f" % d["value"], "ms %.2f" % d["ms_per_step"], "differing", (d.get("parity_check") or {}).get("differing"),
              {a: round(b, 2) for a, b in k.items() if isinstance(b, float)}, flush=True)
