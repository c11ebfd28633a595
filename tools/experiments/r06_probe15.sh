#!/bin/bash
# round 6, probe 15: same-box A/B of the in-flight tile rule (tuning build of the CURRENT sources: default = new rule; RRT_LINEAR16_CFG=6,1,768 = the old shape at N = 9000)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() {  # label, lib, args, env...
  lbl=$1; lib=$2; a=$3; shift 3
  ( for kv in "$@"; do export "$kv"; done
    [ -n "$lib" ] && export RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
}
: > $OUT/r06_p15_ab.txt
for rep in 1 2 3; do
  run "bf16 product" "" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p15_ab.txt
  run "bf16 tune-new" tune "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p15_ab.txt
  run "bf16 tune-old" tune "--dtype bf16 --steps 30 --warmup 5" RRT_LINEAR16_CFG=6,1,768 >> $OUT/r06_p15_ab.txt
done
cat $OUT/r06_p15_ab.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_x
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/p.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "linear_ws" | cut -c1-60,96-140
