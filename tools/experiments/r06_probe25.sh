#!/bin/bash
# round 6, probe 25: XCD-aware block order of the weight-gradient product: tests, dW timing A/B, training step A/B (tuning build)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
timeout 900 python -m pytest tests -m gpu -x -q -k "linear_backward or encoder_backward or gradients or training or train_mode" 2>&1 | tail -3 > $OUT/r06_p25_tests.txt; cat $OUT/r06_p25_tests.txt
: > $OUT/r06_p25_ab.txt
for rep in 1 2; do
  echo "== xcd order" >> $OUT/r06_p25_ab.txt
  timeout 200 python tools/experiments/r06_probe25.py >> $OUT/r06_p25_ab.txt 2>&1
  timeout 200 python tools/prof_train.py 9000 40 >> $OUT/r06_p25_ab.txt 2>&1
  echo "== plain grid" >> $OUT/r06_p25_ab.txt
  RRT_TN_PLAIN_GRID=1 timeout 200 python tools/experiments/r06_probe25.py >> $OUT/r06_p25_ab.txt 2>&1
  RRT_TN_PLAIN_GRID=1 timeout 200 python tools/prof_train.py 9000 40 >> $OUT/r06_p25_ab.txt 2>&1
done
cat $OUT/r06_p25_ab.txt
