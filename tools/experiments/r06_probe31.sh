#!/bin/bash
# round 6, probe 31: how many split-K chunks for the weight-gradient product (RRT_TN_BLOCKS = target block count, tuning build)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
: > $OUT/r06_p31.txt
for tb in 384 512 768 1024 1536 2304; do
  echo "== RRT_TN_BLOCKS=$tb" >> $OUT/r06_p31.txt
  RRT_TN_BLOCKS=$tb timeout 200 python tools/experiments/r06_probe25.py 2>&1 | grep "^dW" >> $OUT/r06_p31.txt
done
cat $OUT/r06_p31.txt
