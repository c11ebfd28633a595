#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp10.txt; : > $T
python -m pytest tests -m gpu -x -q -k "attn_pool or forward_bags or executor" 2>&1 | grep -E "^E|passed|failed" | head -10 | tee -a $T
export GPU_MAX_HW_QUEUES=16
for S in 2 4; do
  echo -n "bags nb=256 S=$S torch-pool streams: " | tee -a $T
  timeout 300 python tools/bench_bags.py uniform 256 $S 2>/dev/null | tail -1 | tee -a $T
  echo -n "bags nb=256 S=$S own streams: " | tee -a $T
  RRT_EXEC_OWN_STREAMS=1 timeout 300 python tools/bench_bags.py uniform 256 $S 2>/dev/null | tail -1 | tee -a $T
done
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
for NB in 64 256; do
  echo -n "bags nb=$NB S=4 torch-pool + stagger: " | tee -a $T
  RRT_EXEC_STAGGER=1 timeout 300 python tools/bench_bags.py uniform $NB 4 2>/dev/null | tail -1 | tee -a $T
  echo -n "bags nb=$NB S=4 own + stagger: " | tee -a $T
  RRT_EXEC_OWN_STREAMS=1 RRT_EXEC_STAGGER=1 timeout 300 python tools/bench_bags.py uniform $NB 4 2>/dev/null | tail -1 | tee -a $T
done
echo -n "mix S=3 torch-pool: " | tee -a $T; timeout 300 python tools/bench_bags.py mix 3 2>/dev/null | tail -1 | tee -a $T
echo -n "mix S=4 torch-pool: " | tee -a $T; timeout 300 python tools/bench_bags.py mix 4 2>/dev/null | tail -1 | tee -a $T
