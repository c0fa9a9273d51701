#!/usr/bin/env python3
"""A/B of an environment switch on the bench: tools/ab_env.py VAR v1,v2[,...] REPEATS [bench flags...]
Runs bench.py once per value and repeat (child processes, one at a time) and prints value / ms / per-kernel times."""
import json, os, subprocess, sys
var, values, repeats = sys.argv[1], sys.argv[2].split(","), int(sys.argv[3])
flags = sys.argv[4:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(repeats):
    for v in values:
        env = dict(os.environ, **{var: v})
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + flags, env=env, capture_output=True, text=True, timeout=400)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(out.returncode)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = (d.get("roofline") or {}).get("kernel_ms_by_kernel") or {}
        print(var, v, "Mrays/s %.1f" % d["value"], "ms %.2f" % d["ms_per_step"], "differing", (d.get("parity_check") or {}).get("differing"),
              {a: round(b, 2) for a, b in k.items() if isinstance(b, float)}, flush=True)
