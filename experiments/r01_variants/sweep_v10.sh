V=""
for t in "2 3 4" "2 4 5" "1 1 1" "2 2 2" "3 2 2" "2 2 4" "1 2 2" "4 3 3"; do set -- $t; V="$V $((10 | (48 << 8) | ($1 << 16) | ($2 << 20) | ($3 << 24)))"; done
bash tools/bench_variants.sh 64 $V
