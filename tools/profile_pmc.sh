#!/bin/bash
# PMC passes for the render megakernel, one counter group per run (rocprofv3
# does not split counters over passes).  Usage: tools/profile_pmc.sh OUTDIR [bench args...]
# Runs a short bench (one 16-spp 1080p launch by default) per pass.  PASSES="tcc1 tcc2" restricts the passes.
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${*:-"--steps 1 --warmup 0 --spp 16 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pass() {
  name=$1; shift
  if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $name "; then return 0; fi
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
pass sq2 SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
pass tcc1 FETCH_SIZE TCC_HIT_sum || exit 1
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum || exit 1
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum || exit 1
pass tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum || exit 1
echo done
