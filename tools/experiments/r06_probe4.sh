#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; rm -f $OUT/r06_p4_lines.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "crmsa or logits or combine or feeder or config4 or forward_bags" 2>&1 | tail -8 > $OUT/r06_p4_tests.txt; cat $OUT/r06_p4_tests.txt
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" >> $OUT/r06_p4_lines.txt; }
for rep in 1 2; do
  for old in 0 1; do
    E=""; [ $old = 1 ] && E="RRT_CRMSA_OLD_PAIR=1"
    env $E RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so timeout 300 python bench.py --config 3 --steps 20 $X 2>/dev/null | line "c3 old_pair=$old"
    env $E RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so timeout 300 python bench.py --config 4 --steps 30 $X 2>/dev/null | line "c4 old_pair=$old"
  done
done
cat $OUT/r06_p4_lines.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o p -- python $R/bench.py --config 3 --streams 1 --steps 6 --raw-loop $X > /tmp/p_c3.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c3/p_results.db > $OUT/r06_p4_c3_1stream.kernel_stats.txt; head -9 $OUT/r06_p4_c3_1stream.kernel_stats.txt | cut -c1-150
rm -rf /tmp/prof_c4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o p -- python $R/bench.py --config 4 --streams 1 --steps 4 $X > /tmp/p_c4.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c4/p_results.db > $OUT/r06_p4_c4_1stream.kernel_stats.txt; grep "crmsa_" $OUT/r06_p4_c4_1stream.kernel_stats.txt | cut -c1-150
