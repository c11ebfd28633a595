#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp7.txt; : > $T
X="--no-cpu-baseline --no-extras"
python -m pytest tests -m gpu -x -q -k "batch_gt_1 or input_ranks or crmsa_stages or test_encoder_matches_reference or attention" 2>&1 | tail -4 | tee -a $T
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for Q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --streams 4 $X > /tmp/b.json 2>/tmp/b.err; line "bench S=4 Q=$Q" /tmp/b.json | tee -a $T
  GPU_MAX_HW_QUEUES=$Q RRT_BENCH_NO_NULL=1 timeout 200 python bench.py --streams 4 $X > /tmp/b.json 2>/tmp/b.err; line "bench S=4 Q=$Q no-null-stream" /tmp/b.json | tee -a $T
done
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
for Q in 8 16; do for M in n p h d; do for S in 2 4; do
  echo -n "bags Q=$Q mode=$M S=$S: " | tee -a $T
  GPU_MAX_HW_QUEUES=$Q RRT_EXEC_STREAMS=$M timeout 200 python tools/bench_bags.py uniform 64 $S 2>/dev/null | tail -1 | tee -a $T
done; done; done
