#!/bin/bash
# Replicates of the best variant-8 settings against variant 5 (64 spp), then the other BASELINE configs.
set -e
for rep in 1 2 3; do
  bash tools/bench_variants.sh 64 5 "8|(16<<8)|(1<<16)|(2<<20)|(3<<24)" "8|(16<<8)|(1<<16)|(3<<20)|(6<<24)" "8|(3<<28)|(16<<8)|(1<<16)|(2<<20)|(3<<24)" "8|(3<<28)|(20<<8)|(1<<16)|(3<<20)|(6<<24)" "8|(20<<8)|(1<<16)|(2<<20)|(3<<24)"
done
for v in 5 "8|(1<<16)|(2<<20)|(3<<24)" "8|(16<<8)|(1<<16)|(2<<20)|(3<<24)"; do
  n=$(python -c "print($v)")
  echo "== configs, variant $v"
  RT_VARIANT=$n timeout -k 10 300 python -u tools/bench_configs.py 2>/dev/null | grep -o "^C[0-9]\|[0-9]* Mrays/s" | paste - -
done
