#!/bin/bash
# round 6, probe 28: parameter-gradient reductions deferred to one launch at the end of the backward: tests (every training /
# gradient / ablation test), training step A/B against RRT_NO_DEFER_REDUCE=1 (tuning build), timeline
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
timeout 1200 python -m pytest tests -m gpu -x -q -k "backward or gradients or training or train or dropout or drop_path or grad or ffn or peg or ablation or autocast" 2>&1 | tail -3 > $OUT/r06_p28_tests.txt; cat $OUT/r06_p28_tests.txt
: > $OUT/r06_p28_ab.txt
for rep in 1 2 3; do
  echo -n "deferred  " >> $OUT/r06_p28_ab.txt; timeout 200 python tools/prof_train.py 9000 80 2>&1 | grep "train step" >> $OUT/r06_p28_ab.txt
  echo -n "immediate " >> $OUT/r06_p28_ab.txt; RRT_NO_DEFER_REDUCE=1 timeout 200 python tools/prof_train.py 9000 80 2>&1 | grep "train step" >> $OUT/r06_p28_ab.txt
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_t
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/tools/prof_train.py 9000 20 > /tmp/prof_t.log 2>&1
python $R/tools/rocprof_timeline.py /tmp/prof_t/t_results.db 50 0.6 > $OUT/r06_p28_train_timeline.txt 2>&1
cat $OUT/r06_p28_ab.txt; cut -c1-100 $OUT/r06_p28_train_timeline.txt | head -56
