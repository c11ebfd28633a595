#!/bin/bash
# round 6, probe 14: the product build with the in-flight tile rule of the 16-bit out-projection: tests + lines
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "linear16 or two_bags or forward_bags or autocast or golden or mil or soak or config4" 2>&1 | tail -3 > $OUT/r06_p14_tests.txt; cat $OUT/r06_p14_tests.txt
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))"; }
cd /tmp; export TMPDIR=/tmp
: > $OUT/r06_p14_lines.txt
for rep in 1 2; do
  timeout 300 python $R/bench.py --dtype bf16 --steps 30 $X 2>/dev/null | line bf16 >> $OUT/r06_p14_lines.txt
  timeout 300 python $R/bench.py --config 2 --steps 100 $X 2>/dev/null | line c2 >> $OUT/r06_p14_lines.txt
  timeout 300 python $R/bench.py --config 3 --steps 40 $X 2>/dev/null | line c3 >> $OUT/r06_p14_lines.txt
  timeout 300 python $R/bench.py --config 4 --steps 30 $X 2>/dev/null | line c4 >> $OUT/r06_p14_lines.txt
done
cat $OUT/r06_p14_lines.txt
