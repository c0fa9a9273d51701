#!/bin/bash
# Round profile of the default render kernel: rocprofv3 kernel trace + stats of the
# bench command, then the PMC passes (tools/profile_pmc.sh).  Usage: tools/profile_round.sh OUTDIR
set -e
OUT=${1:-gpurun_out/round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
echo "trace rc=$?"
bash tools/profile_pmc.sh "$OUT/pmc"
