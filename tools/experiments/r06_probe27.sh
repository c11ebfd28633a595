#!/bin/bash
# round 6, probe 27: CR-MSA's projection with dropout on the split-K small-M kernel (training forward): tests + training step A/B
# (RRT_NO_SPLITK=1 sends every small-M product of the step back to the generic kernel; tuning build)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
timeout 900 python -m pytest tests -m gpu -x -q -k "linear_backward or encoder_backward or gradients or training or train_mode or dropout or drop_path" 2>&1 | tail -3 > $OUT/r06_p27_tests.txt; cat $OUT/r06_p27_tests.txt
: > $OUT/r06_p27_ab.txt
for rep in 1 2 3; do
  timeout 200 python tools/prof_train.py 9000 60 2>&1 | grep "train step" >> $OUT/r06_p27_ab.txt
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_t
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/tools/prof_train.py 9000 20 > /tmp/prof_t.log 2>&1
python $R/tools/rocprof_timeline.py /tmp/prof_t/t_results.db 50 0.6 > $OUT/r06_p27_train_timeline.txt 2>&1
cat $OUT/r06_p27_ab.txt; cut -c1-100 $OUT/r06_p27_train_timeline.txt | head -60
