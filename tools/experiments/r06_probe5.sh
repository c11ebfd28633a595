#!/bin/bash
# round 6, probe 5: crmsa_stream4 (one pass over x1 above 144 tokens per region): stage + whole-path tests, config 3 / 4 lines
# against RRT_NO_CRMSA_STREAM4=1 (tuning switch: needs a tuning build -> here the A/B is new build vs the numbers of r06a), kernel table
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "pair16 or fused16 or crmsa or golden or config4 or autocast or bags_in_flight" 2>&1 | tail -6 > $OUT/r06_p5_tests.txt; cat $OUT/r06_p5_tests.txt
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('one_bag_in_flight') or {}).get('ms_per_bag'), json.dumps(r.get('whole_path'))[:400])"; }
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python $R/bench.py --config 3 --steps 60 $X 2>/dev/null | line c3 >> $OUT/r06_p5_lines.txt
  timeout 300 python $R/bench.py --config 4 --steps 30 $X 2>/dev/null | line c4 >> $OUT/r06_p5_lines.txt
done
cat $OUT/r06_p5_lines.txt
rm -rf /tmp/prof_c3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o p -- python $R/bench.py --config 3 --streams 1 --steps 40 $X > /tmp/p.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c3/p_results.db > $OUT/r06_p5_c3_1stream.kernel_stats.txt; head -12 $OUT/r06_p5_c3_1stream.kernel_stats.txt | cut -c1-150
rm -rf /tmp/prof_c4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o p -- python $R/bench.py --config 4 --streams 1 --steps 10 $X > /tmp/p.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c4/p_results.db > $OUT/r06_p5_c4_1stream.kernel_stats.txt; head -30 $OUT/r06_p5_c4_1stream.kernel_stats.txt | cut -c1-150
