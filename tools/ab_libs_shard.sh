#!/bin/bash
# A/B of library builds with tools/bench_shard.py (kernel 9 on 1/8 and the whole of the bench frame): tools/ab_libs_shard.sh REPEATS lib1.so lib2.so ...
reps=$1; shift
for r in $(seq 1 "$reps"); do
  for lib in "$@"; do
    cp "$lib" raytracing_amd/librt_hip.so || exit 1
    echo "== $lib"
    RT_KERNELS=9 RT_STRIDES=${RT_STRIDES:-8,1} timeout -k 10 200 python tools/bench_shard.py 2>&1 | grep stride || exit 1
  done
done
