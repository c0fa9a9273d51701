cp experiments/ab_libs/persist_w512.so raytracing_amd/librt_hip.so
export RT_WF_PERSIST=1 RT_STRIDES=8,1
for k in 4105 6153 8201 12297; do echo "== regen knob variant $k"; RT_KERNELS=$k timeout -k 10 100 python tools/bench_shard.py 2>&1 | grep stride || exit 1; done
for s in 8 16 48 64; do echo "== suspend $s"; RT_WF_TUNING=$s,-1,-1,-1 RT_KERNELS=9 timeout -k 10 100 python tools/bench_shard.py 2>&1 | grep stride || exit 1; done
