#!/bin/bash
# round 6, probe 18: CR-MSA's front as one block per region when several fp32 bags are in flight (crmsa_region_kernel<k>): tests + same-box A/B (tuning build)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "crmsa or two_bags or forward_bags or executor or golden or soak or config4 or bags" 2>&1 | tail -3 > $OUT/r06_p18_tests.txt; cat $OUT/r06_p18_tests.txt
run() {  # label, args, env...
  lbl=$1; a=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
}
: > $OUT/r06_p18_ab.txt
for rep in 1 2 3; do
  run "f32 region1(new)" "--dtype f32 --steps 8 --warmup 3" >> $OUT/r06_p18_ab.txt
  run "f32 region1(round-1 kernel)" "--dtype f32 --steps 8 --warmup 3" RRT_NO_REGION_GPR=1 >> $OUT/r06_p18_ab.txt
  run "f32 region4" "--dtype f32 --steps 8 --warmup 3" RRT_NO_REGION_INFLIGHT=1 >> $OUT/r06_p18_ab.txt
  run "bf16 region4" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p18_ab.txt
  run "bf16 region1(new)" "--dtype bf16 --steps 30 --warmup 5" RRT_REGION_INFLIGHT_LOWP=1 >> $OUT/r06_p18_ab.txt
done
cat $OUT/r06_p18_ab.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_x
RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype f32 --streams 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > /tmp/p.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "crmsa_" | cut -c1-60,96-140
