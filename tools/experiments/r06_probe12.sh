#!/bin/bash
# round 6, probe 12 (tuning build): tile shape x resident blocks of the 16-bit out-projection WITH SEVERAL BAGS IN FLIGHT (the round-3 rule was swept with one)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
run() {  # label, bench args (quoted), env assignments...
  lbl=$1; a=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
}
: > $OUT/r06_p12_ab.txt
for rep in 1 2; do
  run "c1 base" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p12_ab.txt
  for c in 9,1,512 9,1,256 8,1,512 9,2,256 9,2,512 8,2,256 6,2,384 6,2,512 5,2,512; do
    run "c1 $c" "--dtype bf16 --steps 30 --warmup 5" RRT_LINEAR16_CFG=$c >> $OUT/r06_p12_ab.txt
  done
  run "c3 base" "--config 3 --steps 40 --warmup 5" >> $OUT/r06_p12_ab.txt
  for c in 9,2,512 8,2,512 9,2,256 8,2,256 9,1,512 8,1,256 6,2,512; do
    run "c3 $c" "--config 3 --steps 40 --warmup 5" RRT_LINEAR16_CFG=$c >> $OUT/r06_p12_ab.txt
  done
  run "c4 base" "--config 4 --steps 30 --warmup 5" >> $OUT/r06_p12_ab.txt
  for c in 9,1,512 9,2,256 8,2,256 6,2,512; do
    run "c4 $c" "--config 4 --steps 30 --warmup 5" RRT_LINEAR16_CFG=$c >> $OUT/r06_p12_ab.txt
  done
done
cat $OUT/r06_p12_ab.txt
