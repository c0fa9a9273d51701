#!/bin/bash
# Variant 8, second pass: idle x weights x join threshold code around the best of sweep_v8.sh.
set -e
spp=${1:-64}
vs=()
for idle in 8 12 16 20; do
  for w in "1 2 3" "1 3 6" "2 5 6"; do
    set -- $w
    vs+=("8|($idle<<8)|($1<<16)|($2<<20)|($3<<24)")
  done
done
for j in 2 3 4 5; do vs+=("8|($j<<28)|(16<<8)|(1<<16)|(2<<20)|(3<<24)"); done
bash tools/bench_variants.sh $spp 5 "${vs[@]}" 5
