#!/bin/bash
# round 6, probe 8: weight gradients on a side stream (LinBwdFork): gradient tests, A/B of the training step (RRT_BWD_NO_FORK=1)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "backward or gradients or train or autocast or mil" 2>&1 | tail -4 > $OUT/r06_p8_tests.txt; cat $OUT/r06_p8_tests.txt
: > $OUT/r06_p8_train.txt
for rep in 1 2; do
  for nf in 0 1; do
    if [ $nf = 1 ]; then export RRT_BWD_NO_FORK=1; else unset RRT_BWD_NO_FORK; fi
    echo "no_fork=$nf: $(timeout 300 python $R/tools/bench_train.py 9000 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT/r06_p8_train.txt
  done
done
unset RRT_BWD_NO_FORK
cat $OUT/r06_p8_train.txt
