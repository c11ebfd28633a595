#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp14.txt; : > $T
export GPU_MAX_HW_QUEUES=16
for V in "RRT_EXEC_POOL_OFFSET=0" "RRT_EXEC_POOL_OFFSET=1" "RRT_EXEC_POOL_OFFSET=2" "RRT_EXEC_POOL_OFFSET=3" "RRT_EXEC_OWN_STREAMS=1" "GPU_MAX_HW_QUEUES=4 RRT_EXEC_POOL_OFFSET=0" "GPU_MAX_HW_QUEUES=8 RRT_EXEC_POOL_OFFSET=0" "BAGS_CALLER_STREAM=1"; do
  echo -n "bags nb=256 S=4 [$V]: " | tee -a $T
  env $V timeout 300 python tools/bench_bags.py uniform 256 4 2>/dev/null | tail -1 | tee -a $T
done
