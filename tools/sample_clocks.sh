#!/bin/bash
# Shader clock and power while the bench runs: samples rocm-smi every 0.2 s during `python bench.py` (N=1, short).
# Usage: tools/sample_clocks.sh [OUT]
OUT=${1:-gpurun_out/clocks.txt}
( for i in $(seq 1 150); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > "$OUT" &
S=$!
timeout -k 10 200 python bench.py --no-cpu-baseline --no-probe --no-wang --no-frame1024 --steps 20 > /dev/null 2>&1
kill $S 2>/dev/null
wait $S 2>/dev/null
sort "$OUT" | uniq -c | sort -rn | head -12
