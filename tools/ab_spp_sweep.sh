#!/bin/bash
# A/B of library builds over samples per launch (tools/bench_spp_sweep.py, kernel 9 column) on the dragon and teapot scenes:
# tools/ab_spp_sweep.sh lib1.so lib2.so ...   The last build stays in place.
for lib in "$@"; do
  cp "$lib" raytracing_amd/librt_hip.so || exit 1
  echo "== $lib"
  RT_SCENE=dragon RT_SPPS=1,4,16,64,256 timeout -k 10 200 python tools/bench_spp_sweep.py 2>&1 | grep spp | sed 's/kernel 8:.*| //' || exit 1
  RT_SCENE=teapot RT_SPPS=1,16,64,256 timeout -k 10 200 python tools/bench_spp_sweep.py 2>&1 | grep spp | sed 's/kernel 8:.*| //' || exit 1
done
