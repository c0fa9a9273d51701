for c in 0 1 3 5 6; do
  echo "== cfg $c"
  RT_WF_TRACE_CFG=$c RT_KERNELS=9 RT_STRIDES=8,1 timeout -k 10 200 python tools/bench_shard.py || exit 1
  RT_WF_TRACE_CFG=$c RT_RNG=pcg RT_SPP=256 RT_VARIANT=9 timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep -E "^C|Mrays|\|" || exit 1
done
