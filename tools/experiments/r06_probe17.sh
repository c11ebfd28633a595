#!/bin/bash
# round 6, probe 17 (tuning build): CR-MSA's front as ONE block per region (crmsa_region_kernel: 64 blocks, not chip-wide) with several bags in flight
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() {  # label, args, env...
  lbl=$1; a=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
}
: > $OUT/r06_p17_ab.txt
for rep in 1 2; do
  run "bf16 region4" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p17_ab.txt
  run "bf16 region1" "--dtype bf16 --steps 30 --warmup 5" RRT_NO_CRMSA_REGION4=1 RRT_CRMSA_REGION=1 >> $OUT/r06_p17_ab.txt
  run "f32 region4" "--dtype f32 --steps 8 --warmup 3" >> $OUT/r06_p17_ab.txt
  run "f32 region1" "--dtype f32 --steps 8 --warmup 3" RRT_NO_CRMSA_REGION4=1 RRT_CRMSA_REGION=1 >> $OUT/r06_p17_ab.txt
done
cat $OUT/r06_p17_ab.txt
