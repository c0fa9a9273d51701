#!/bin/bash
# Variant 8: vote weights (pair, leaf, instance), regeneration threshold and join threshold code, 64 spp.
set -e
spp=${1:-64}
vs=()
for w in "2 4 5" "1 2 3" "2 3 5" "2 5 6" "1 3 6" "3 4 6" "1 1 3" "2 2 5" "4 4 5"; do
  set -- $w
  vs+=("8|($1<<16)|($2<<20)|($3<<24)")
done
for idle in 16 24 40 48; do vs+=("8|($idle<<8)"); done
for j in 2 3 4; do vs+=("8|($j<<28)|(24<<8)" "8|($j<<28)|(40<<8)"); done
bash tools/bench_variants.sh $spp 5 "${vs[@]}" 5
