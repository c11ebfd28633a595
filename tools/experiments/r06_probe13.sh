#!/bin/bash
# round 6, probe 13 (tuning build): the 16-bit out-projection's tile shape over BAG SIZES with four bags in flight (bf16, config 1 line at RRT_BENCH_N)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
run() {  # label, env assignments...
  lbl=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py --dtype bf16 --streams 4 --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'])" )
}
: > $OUT/r06_p13_ab.txt
for n in 3000 5000 7000 8000 9000 10500 12000 15000; do
  run "N=$n base" RRT_BENCH_N=$n >> $OUT/r06_p13_ab.txt
  for c in 4,2,512 4,2,1024 5,2,512 6,2,512 6,2,384 8,2,256 9,2,256 9,1,512; do
    run "N=$n $c" RRT_BENCH_N=$n RRT_LINEAR16_CFG=$c >> $OUT/r06_p13_ab.txt
  done
done
cat $OUT/r06_p13_ab.txt
