#!/bin/bash
# usage: tools/bench_variants.sh [spp] variant...   (variants may be expressions, e.g. '5|(8<<8)')
set -e
spp=$1; shift
for v in "$@"; do
  n=$(python -c "print($v)")
  timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp $spp --no-cpu-baseline --variant $n $BENCH_EXTRA 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant $v', 'Mrays/s %.0f'%d['value'], 'kernel_ms %.1f'%d['kernel_ms_per_launch'])
"
done
