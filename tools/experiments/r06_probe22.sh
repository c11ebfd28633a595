#!/bin/bash
# round 6, probe 22 (tuning build): fp32, four bags in flight -- the inner MSA's GEMMs as K-split 16-wave blocks (the one-bag shape) instead of light 4-wave blocks
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
run() { lbl=$1; shift; ( for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py --dtype f32 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'])" ); }
: > $OUT/r06_p22_ab.txt
for rep in 1 2 3; do
  run base >> $OUT/r06_p22_ab.txt
  run inner_solo RRT_INNER_SOLO=1 >> $OUT/r06_p22_ab.txt
  run prio0 RRT_X=1 >> $OUT/r06_p22_ab.txt
done
cat $OUT/r06_p22_ab.txt
