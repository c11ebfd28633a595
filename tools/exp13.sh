#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp13.txt; : > $T
export GPU_MAX_HW_QUEUES=16 RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
for V in "" "RRT_EXEC_NOJOIN=1" "RRT_EXEC_NOFORK=1" "RRT_EXEC_NOFORK=1 RRT_EXEC_NOJOIN=1" "BAGS_CALLER_STREAM=1" "BAGS_CALLER_STREAM=1 RRT_EXEC_NOJOIN=1"; do
  for NB in 64 512; do
    echo -n "bags nb=$NB S=4 [$V]: " | tee -a $T
    env $V timeout 300 python tools/bench_bags.py uniform $NB 4 2>/dev/null | tail -1 | tee -a $T
  done
done
