#!/bin/bash
# Regeneration threshold (idle lanes) on the BASELINE configs other than C3, default vote weights.
for idle in 8 16 24 32 48 56; do
  echo "== idle $idle"
  RT_VARIANT=$((${BASE:-8} | (idle << 8))) timeout -k 10 300 python -u tools/bench_configs.py 2>/dev/null | grep -o "^C[0-9]\|[0-9]* Mrays/s" | paste - -
done
