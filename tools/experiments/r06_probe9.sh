#!/bin/bash
# round 6, probe 9: the classifier's pooling-score product on 16-bit operands (y16 from the encoder's last kernel): tests, config 2 lines, kernel table
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "mil or pool or G8 or G11 or G13 or dispatch or feeder" 2>&1 | tail -4 > $OUT/r06_p9_tests.txt; cat $OUT/r06_p9_tests.txt
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'))"; }
cd /tmp; export TMPDIR=/tmp
: > $OUT/r06_p9_lines.txt
for rep in 1 2; do
  timeout 300 python $R/bench.py --config 2 --steps 100 $X 2>/dev/null | line c2 >> $OUT/r06_p9_lines.txt
done
cat $OUT/r06_p9_lines.txt
rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o p -- python $R/bench.py --config 2 --streams 1 --steps 40 $X > /tmp/p.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c2/p_results.db > $OUT/r06_p9_c2_1stream.kernel_stats.txt; head -18 $OUT/r06_p9_c2_1stream.kernel_stats.txt | cut -c1-150
