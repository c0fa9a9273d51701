#!/bin/bash
# Sweep of the vote weights (leaf, instance) and the regeneration threshold of the default kernel on C3, 64 spp.
set -e
V=""
for idle in 12 16 24; do for wl in 2 3 4 5; do for wi in 4 6 10; do
  V="$V $((5 | (idle << 8) | (1 << 16) | (wl << 20) | (wi << 24)))"
done; done; done
bash tools/bench_variants.sh 64 $V
