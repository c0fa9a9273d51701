#!/bin/bash
# A/B of library variants: experiments/r03_window/run.sh "256 512 1024" [bench flags]; swaps the variant in (on the box's scratch copy)
set -e
VARIANTS=$1; shift
cp raytracing_amd/librt_hip.so /tmp/librt_hip_orig.so
for rep in 1 2; do for v in $VARIANTS; do
  cp experiments/r03_window/librt_hip_$v.so raytracing_amd/librt_hip.so
  timeout -k 10 300 python tools/ab_env.py RT_VARIANT_TAG $v 1 "$@"
done; done
cp /tmp/librt_hip_orig.so raytracing_amd/librt_hip.so
