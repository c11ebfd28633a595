#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp12.txt; : > $T
python -m pytest tests -m gpu -x -q -k "forward_bags or executor or cache or concurrent or attn_pool" 2>&1 | grep -E "^E|passed|failed" | head -10 | tee -a $T
export GPU_MAX_HW_QUEUES=16
for S in 1 2 3 4; do for NB in 64 256; do
  echo -n "bags nb=$NB S=$S: " | tee -a $T
  timeout 300 python tools/bench_bags.py uniform $NB $S 2>/dev/null | tail -1 | tee -a $T
done; done
for S in 2 3 4; do echo -n "mix S=$S: " | tee -a $T; timeout 300 python tools/bench_bags.py mix $S 2>/dev/null | tail -1 | tee -a $T; done
for S in 3 4; do
timeout 200 python bench.py --config 4 --streams $S --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 S=$S', r['value'])" | tee -a $T
done
