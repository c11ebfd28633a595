#!/bin/bash
# round 6, probe 20 (tuning build): tile shape of the classifier's patch_to_emb product (K = 1024) with four slides in flight (configs[2])
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
run() { lbl=$1; shift; ( for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py --config 2 --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'])" ); }
: > $OUT/r06_p20_ab.txt
for rep in 1 2; do
  run base >> $OUT/r06_p20_ab.txt
  for c in 6,2,512 5,2,512 9,2,256 8,2,256 9,1,512 6,1,1024 4,2,512; do run "kbig $c" RRT_LINEAR16_CFG_KBIG=$c >> $OUT/r06_p20_ab.txt; done
done
cat $OUT/r06_p20_ab.txt
