#!/bin/bash
# round 6, probe 16: crmsa_region4 with gamma . phi in registers + wave_sum4 (k = 1, 3, 5): tests, same-box A/B (tuning build: RRT_NO_REGION4_GPR=1 = the old kernel), kernel time
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "crmsa or two_bags or forward_bags or golden or oracle_f64 or soak or config4 or autocast" 2>&1 | tail -3 > $OUT/r06_p16_tests.txt; cat $OUT/r06_p16_tests.txt
run() {  # label, args, env...
  lbl=$1; a=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
}
: > $OUT/r06_p16_ab.txt
for rep in 1 2 3; do
  run "bf16 gpr" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p16_ab.txt
  run "bf16 old" "--dtype bf16 --steps 30 --warmup 5" RRT_NO_REGION4_GPR=1 >> $OUT/r06_p16_ab.txt
  run "c4 gpr" "--config 4 --steps 30 --warmup 5" >> $OUT/r06_p16_ab.txt
  run "c4 old" "--config 4 --steps 30 --warmup 5" RRT_NO_REGION4_GPR=1 >> $OUT/r06_p16_ab.txt
  run "f32 gpr" "--dtype f32 --steps 8 --warmup 3" >> $OUT/r06_p16_ab.txt
  run "f32 old" "--dtype f32 --steps 8 --warmup 3" RRT_NO_REGION4_GPR=1 >> $OUT/r06_p16_ab.txt
done
cat $OUT/r06_p16_ab.txt
cd /tmp; export TMPDIR=/tmp
for v in gpr old; do
  rm -rf /tmp/prof_x; ( [ $v = old ] && export RRT_NO_REGION4_GPR=1; RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype bf16 --streams 1 --steps 40 --no-cpu-baseline --no-extras > /tmp/p.log 2>&1 )
  python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "crmsa_region4" | cut -c1-60,96-140 | sed "s/^/$v /"
done
