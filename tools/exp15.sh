#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp15.txt; : > $T
export GPU_MAX_HW_QUEUES=16
for V in "RRT_BAGS_NULL_MODE=hostsync" "RRT_BAGS_NULL_MODE=async" "RRT_BAGS_NULL_MODE=direct"; do
 for NB in 16 64 256; do
  echo -n "bags nb=$NB S=4 [$V]: " | tee -a $T
  env $V timeout 300 python tools/bench_bags.py uniform $NB 4 2>/dev/null | tail -1 | tee -a $T
 done
done
for S in 2 3; do echo -n "bags nb=64 S=$S hostsync: " | tee -a $T; timeout 300 python tools/bench_bags.py uniform 64 $S 2>/dev/null | tail -1 | tee -a $T; done
for S in 3 4; do echo -n "mix S=$S hostsync: " | tee -a $T; timeout 300 python tools/bench_bags.py mix $S 2>/dev/null | tail -1 | tee -a $T; done
python -m pytest tests -m gpu -x -q -k "forward_bags or executor or cache" 2>&1 | grep -E "^E|passed|failed" | head -5 | tee -a $T
