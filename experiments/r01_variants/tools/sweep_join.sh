#!/bin/bash
# Variant 8 (vote scheduling with shared fetches): join threshold codes 1..7 (1, 4, 6, 12, 16, 24 lanes, never) and the default.
set -e
spp=${1:-64}
bash tools/bench_variants.sh $spp 5 8 '8|(1<<28)' '8|(2<<28)' '8|(3<<28)' '8|(4<<28)' '8|(5<<28)' '8|(6<<28)' '8|(7<<28)' 5
