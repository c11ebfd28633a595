#!/bin/bash
# round 6, probe 30: CR-MSA backward's front (tokdot + wsum + dropout mask) as one region-block launch: tests, training step A/B
# against RRT_NO_BWD_FRONT=1 (tuning build), timeline.  The kernel under test (crmsa_bwd_front_kernel<KM>, one 16-wave block per
# region) was NOT kept: 16.3 us against 19.4 us for the three launches it replaced and no change in the step
# (profiles/r06_crmsa_bwd_front_ab.txt); with the tree's sources both legs run the three launches.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
timeout 1200 python -m pytest tests -m gpu -x -q -k "backward or gradients or training or train or dropout or drop_path or grad or ffn or peg or ablation or autocast" 2>&1 | tail -3 > $OUT/r06_p30_tests.txt; cat $OUT/r06_p30_tests.txt
: > $OUT/r06_p30_ab.txt
for rep in 1 2 3; do
  echo -n "front   " >> $OUT/r06_p30_ab.txt; timeout 200 python tools/prof_train.py 9000 80 2>&1 | grep "train step" >> $OUT/r06_p30_ab.txt
  echo -n "3 stages " >> $OUT/r06_p30_ab.txt; RRT_NO_BWD_FRONT=1 timeout 200 python tools/prof_train.py 9000 80 2>&1 | grep "train step" >> $OUT/r06_p30_ab.txt
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_t
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/tools/prof_train.py 9000 20 > /tmp/prof_t.log 2>&1
python $R/tools/rocprof_timeline.py /tmp/prof_t/t_results.db 50 0.6 > $OUT/r06_p30_train_timeline.txt 2>&1
cat $OUT/r06_p30_ab.txt; cut -c1-100 $OUT/r06_p30_train_timeline.txt | grep -i "front\|tokdot\|wsum\|drop_mask\|ln_backward"
