#!/bin/bash
# SQ instruction / wait counters of two kernel variants on one 16-spp launch (separate rocprofv3 passes).
set -e
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  n=$(python3 -c "print($v)")
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$n
  rm -rf $out
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $out -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp 16 --no-cpu-baseline --variant $n > /dev/null 2>&1
  echo "== variant $v"
  python3 - $out <<'PY'
import sys,glob,csv,collections
tot=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_render' in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value'])
for k in sorted(tot): print(k, '%.4g'%tot[k])
PY
done
