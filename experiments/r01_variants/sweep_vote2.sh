#!/bin/bash
# Finer sweep of the pair : leaf vote-weight ratio of the default kernel on C3, 64 spp (regeneration threshold 16).
set -e
V=""
for t in "1 1 6" "4 5 15" "3 4 15" "2 3 12" "3 5 15" "4 7 15" "1 2 6" "2 5 12" "1 2 8" "2 3 8" "1 2 6"; do
  set -- $t
  V="$V $((5 | (16 << 8) | ($1 << 16) | ($2 << 20) | ($3 << 24)))"
done
bash tools/bench_variants.sh 64 $V
