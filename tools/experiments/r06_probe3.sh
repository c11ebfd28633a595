#!/bin/bash
# round 6, third probe: the streaming CR-MSA pair for large regions (crmsa_logits512 / crmsa_combine512): tests, A/B, kernel table
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "crmsa or logits or combine or golden or config4 or G4 or G16 or G19" 2>&1 | tail -4 > $OUT/r06_p3_tests.txt; cat $OUT/r06_p3_tests.txt
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" >> $OUT/r06_p3_lines.txt; }
for rep in 1 2; do
  for old in 0 1; do
    E=""; [ $old = 1 ] && E="RRT_CRMSA_OLD_PAIR=1"
    env $E RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so timeout 300 python bench.py --config 3 --steps 20 $X 2>/dev/null | line "c3 old_pair=$old"
    env $E RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so timeout 300 python bench.py --config 4 --steps 30 $X 2>/dev/null | line "c4 old_pair=$old"
  done
done
timeout 300 python bench.py --dtype f32 --steps 20 --warmup 5 $X 2>/dev/null | line "f32 R=64 driver-steps"
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 $X 2>/dev/null | line "bf16 R=64 driver-steps"
cat $OUT/r06_p3_lines.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o p -- python $R/bench.py --config 3 --streams 1 --steps 6 --raw-loop $X > /tmp/p_c3.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c3/p_results.db > $OUT/r06_p3_c3_1stream.kernel_stats.txt; head -12 $OUT/r06_p3_c3_1stream.kernel_stats.txt | cut -c1-150
cd $R; tools/repro_packed_fp32.sh 200 40
