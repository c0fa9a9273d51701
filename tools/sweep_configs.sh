#!/bin/bash
# Vote weights (pair leaf inst) on the BASELINE configs other than C3 (tools/bench_configs.py).
for t in "1 2 3" "1 2 2" "1 2 1" "2 4 5" "1 2 3"; do
  set -- $t
  echo "== weights $t"
  RT_VARIANT=$((8 | ($1 << 16) | ($2 << 20) | ($3 << 24))) timeout -k 10 300 python -u tools/bench_configs.py 2>/dev/null | grep -o "^C[0-9]\|[0-9]* Mrays/s" | paste - - 
done
