#!/bin/bash
# round 6, probe 6: the merged 16-bit launch (rmsa_pair16 PROJ): bit-identity tests, config 3 lines, kernel table
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "${1:-pair16}" 2>&1 | tail -6 > $OUT/r06_p6_tests.txt; cat $OUT/r06_p6_tests.txt
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))"; }
cd /tmp; export TMPDIR=/tmp
: > $OUT/r06_p6_lines.txt
for rep in 1 2; do
  timeout 300 python $R/bench.py --config 3 --steps 60 $X 2>/dev/null | line c3 >> $OUT/r06_p6_lines.txt
done
cat $OUT/r06_p6_lines.txt
rm -rf /tmp/prof_c3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o p -- python $R/bench.py --config 3 --streams 1 --steps 40 $X > /tmp/p.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c3/p_results.db > $OUT/r06_p6_c3_1stream.kernel_stats.txt; head -9 $OUT/r06_p6_c3_1stream.kernel_stats.txt | cut -c1-150
