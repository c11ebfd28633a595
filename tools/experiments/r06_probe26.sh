#!/bin/bash
# round 6, probe 26: one training step as a timeline (rocprofv3 --kernel-trace): every dispatch with start / end, to see the gaps
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/tools/prof_train.py 9000 20 > /tmp/prof_t.log 2>&1
grep "train step" /tmp/prof_t.log
python $R/tools/rocprof_timeline.py /tmp/prof_t/t_results.db 70 0.6 > $OUT/r06_p26_train_timeline.txt 2>&1
cut -c1-170 $OUT/r06_p26_train_timeline.txt
