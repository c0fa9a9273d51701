#!/bin/bash
# Round profile of the default bench command: rocprofv3 kernel trace + stats of `python3 bench.py`, no flags (kernel 9 with the
# keyed PCG RNG, then the megakernel with the reference's RNG), then the PMC passes of one 256-spp launch of each
# (tools/profile_pmc.sh).  Usage: tools/profile_round.sh OUTDIR
set -e
OUT=${1:-gpurun_out/round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py > "$OUT/bench_under_rocprof.log" 2>&1
echo "trace rc=$?"
bash tools/profile_pmc.sh "$OUT/pmc_pcg" --steps 1 --warmup 0 --no-cpu-baseline --no-probe --no-wang --no-frame1024 --rng pcg --variant 9   # kernel 9 named: no trial launches
bash tools/profile_pmc.sh "$OUT/pmc_wang" --steps 1 --warmup 0 --no-cpu-baseline --no-probe --no-frame1024 --rng wang
# the reference's RNG as the reference samples: 256 progressive frames of 1 spp in one batch, through kernel 9 (named: no trials)
bash tools/profile_pmc.sh "$OUT/pmc_wang_frames" --steps 1 --warmup 0 --no-cpu-baseline --no-probe --rng wang --progressive --variant 9
