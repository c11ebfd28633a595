#!/bin/bash
# round 6, probe 19 (tuning build): 256-column tiles for the 16-bit out-projection with four bags in flight
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
run() { lbl=$1; shift; ( for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" ); }
: > $OUT/r06_p19_ab.txt
for rep in 1 2; do
  run base >> $OUT/r06_p19_ab.txt
  for c in 6,4,256 4,4,512 5,4,256 8,4,256 3,4,512 6,2,512 5,2,512; do run "$c" RRT_LINEAR16_CFG=$c >> $OUT/r06_p19_ab.txt; done
done
cat $OUT/r06_p19_ab.txt
