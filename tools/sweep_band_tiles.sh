#!/bin/bash
# Sweep of RT_WF_BAND_TILES (a build with experiments/r03_persistent/band_pieces.patch applied: tiles per piece of the frame that one XCD class of the persistent
# kernel works through): tools/sweep_band_tiles.sh LIB T1 T2 ...
lib=$1; shift
cp "$lib" raytracing_amd/librt_hip.so || exit 1
for t in "$@"; do
  echo "== tiles $t"
  export RT_WF_BAND_TILES=$t
  RT_KERNELS=9 RT_STRIDES=8,1 timeout -k 10 200 python tools/bench_shard.py 2>&1 | grep stride || exit 1
  RT_SCENE=dragon RT_SPPS=1,4,16 timeout -k 10 200 python tools/bench_spp_sweep.py 2>&1 | grep spp | sed 's/kernel 8:.*| //' || exit 1
  RT_SCENE=teapot RT_SPPS=1,16,64,256 timeout -k 10 200 python tools/bench_spp_sweep.py 2>&1 | grep spp | sed 's/kernel 8:.*| //' || exit 1
done
