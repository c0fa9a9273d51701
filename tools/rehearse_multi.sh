#!/bin/bash
# Rehearsal of bench.py's multi-rank code path on a box with ONE GPU: 2 (or $1) ranks share cuda:0 and talk over gloo.
# Checks that every sharding mode runs to its JSON line; the numbers mean nothing.
set -e
N=${1:-2}
for mode in frames rows interleave; do
  echo "== $mode"
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 200)) bench.py --gpus $N --steps 2 --warmup 1 --spp 16 --shard $mode --rehearse --no-frame1024
done
