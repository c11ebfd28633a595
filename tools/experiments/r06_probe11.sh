#!/bin/bash
# round 6, probe 11 (tuning build): can crmsa_region4 and the 16-bit out-projection of DIFFERENT bags share CUs?  bf16, four bags in flight:
# the product's shapes against region4 as eight 4-wave blocks per region (124 VGPRs: one wave per SIMD) and / or the projection at two blocks per CU
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
run() {  # label, env assignments...
  lbl=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
}
: > $OUT/r06_p11_ab.txt
for rep in 1 2; do
  run base >> $OUT/r06_p11_ab.txt
  run r4x8 RRT_REGION4_CFG=8 >> $OUT/r06_p11_ab.txt
  run proj2 RRT_LINEAR16_CFG=6,1,512 >> $OUT/r06_p11_ab.txt
  run r4x8+proj2 RRT_REGION4_CFG=8 RRT_LINEAR16_CFG=6,1,512 >> $OUT/r06_p11_ab.txt
  run proj9x1 RRT_LINEAR16_CFG=9,1,512 >> $OUT/r06_p11_ab.txt
  run proj9x2 RRT_LINEAR16_CFG=9,2,256 >> $OUT/r06_p11_ab.txt
  run r4x8+proj9x2 RRT_REGION4_CFG=8 RRT_LINEAR16_CFG=9,2,256 >> $OUT/r06_p11_ab.txt
  run pair16nh1 RRT_PAIR16_NH=1 >> $OUT/r06_p11_ab.txt
done
cat $OUT/r06_p11_ab.txt
