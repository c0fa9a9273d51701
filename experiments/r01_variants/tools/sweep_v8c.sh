#!/bin/bash
# Variant 8, third pass (after the scalar-lean rewrite): weights, join threshold code and idle threshold, 64 spp.
set -e
spp=${1:-64}
vs=(8)
for w in "1 1 2" "1 2 3" "1 2 4" "1 3 4" "2 3 4" "2 3 6" "1 1 1" "3 4 6"; do
  set -- $w
  vs+=("8|($1<<16)|($2<<20)|($3<<24)")
done
for j in 1 2 3 4 5; do vs+=("8|($j<<28)"); done
for idle in 8 12 20 24 32; do vs+=("8|($idle<<8)"); done
bash tools/bench_variants.sh $spp "${vs[@]}" 8
