#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_libs.sh REPEATS lib1.so lib2.so ... -- [bench.py flags]
# Each build is copied over raytracing_amd/librt_hip.so (the box's tree is a scratch copy) and bench.py runs as a child;
# prints value / ms / parity / per-kernel times per build and repeat.  The last build stays in place.
set -u
reps=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ $# -gt 0 ] && shift
for r in $(seq 1 "$reps"); do
  for lib in "${libs[@]}"; do
    cp "$lib" raytracing_amd/librt_hip.so || exit 1
    timeout -k 10 300 python bench.py "$@" > /tmp/ab_out.json 2>/tmp/ab_err.txt || { tail -5 /tmp/ab_err.txt; exit 1; }
    python - "$lib" <<'P'
import json, sys
d = json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1])
k = (d.get("roofline") or {}).get("kernel_ms_by_kernel") or {}
print(sys.argv[1], "Mrays/s %.1f" % d["value"], "ms %.2f" % d["ms_per_step"], "differing", (d.get("parity_check") or {}).get("differing"),
      {a: round(b, 2) for a, b in k.items() if isinstance(b, float)}, flush=True)
P
  done
done
