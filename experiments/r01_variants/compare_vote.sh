#!/bin/bash
# A/B of vote weights at the bench's own shape (256 spp, 3 timed steps), interleaved repeats.
set -e
for rep in 1 2 3; do
for t in "1 2 4" "2 4 5" "1 2 3"; do
  set -- $t
  v=$((5 | ($1 << 16) | ($2 << 20) | ($3 << 24)))
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --variant $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('weights $t', 'Mrays/s %.0f'%d['value'], 'kernel_ms %.1f'%d['kernel_ms_per_launch'])
"
done; done
