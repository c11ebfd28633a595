#!/bin/bash
# round 6, probe 23: crmsa_region4 without its hand-over tail + the merge as its own launch (16-bit modes, bags in flight): tests, same-box A/B (tuning build)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "two_bags or forward_bags or executor or golden or soak or config4 or autocast or mil or bags" 2>&1 | tail -3 > $OUT/r06_p23_tests.txt; cat $OUT/r06_p23_tests.txt
run() { lbl=$1; a=$2; shift 2; ( for kv in "$@"; do export "$kv"; done
    export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" ); }
: > $OUT/r06_p23_ab.txt
for rep in 1 2 3; do
  run "bf16 split" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p23_ab.txt
  run "bf16 whole" "--dtype bf16 --steps 30 --warmup 5" RRT_NO_REGION4_SPLIT=1 >> $OUT/r06_p23_ab.txt
  run "c2 split" "--config 2 --steps 100 --warmup 5" >> $OUT/r06_p23_ab.txt
  run "c2 whole" "--config 2 --steps 100 --warmup 5" RRT_NO_REGION4_SPLIT=1 >> $OUT/r06_p23_ab.txt
  run "c4 split" "--config 4 --steps 30 --warmup 5" >> $OUT/r06_p23_ab.txt
  run "c4 whole" "--config 4 --steps 30 --warmup 5" RRT_NO_REGION4_SPLIT=1 >> $OUT/r06_p23_ab.txt
done
cat $OUT/r06_p23_ab.txt
