#!/bin/bash
# round 6, probe 29: eight waves per block in ln_backward / crmsa_bwd_dx (rows per wave 4.4 -> 2.2 at N = 9000; dx2 / add rows
# requested with the row): tests, training step A/B against RRT_LNB_NW4=1 RRT_DXB_NW4=1 (tuning build), kernel table.
# The kernels under test (ln_backward_kernel<NV, NW>, crmsa_bwd_dx_kernel<NV, MLP, NW, KM> and their two switches) were NOT kept:
# no gain (profiles/r06_bwd_rows_8waves_ab.txt); with the tree's sources both legs of this script run the same four-wave kernels.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
timeout 1200 python -m pytest tests -m gpu -x -q -k "backward or gradients or training or train or dropout or drop_path or grad or ffn or peg or ablation or autocast or layernorm" 2>&1 | tail -3 > $OUT/r06_p29_tests.txt; cat $OUT/r06_p29_tests.txt
: > $OUT/r06_p29_ab.txt
for rep in 1 2 3; do
  echo -n "8 waves  " >> $OUT/r06_p29_ab.txt; timeout 200 python tools/prof_train.py 9000 80 2>&1 | grep "train step" >> $OUT/r06_p29_ab.txt
  echo -n "4 waves  " >> $OUT/r06_p29_ab.txt; RRT_LNB_NW4=1 RRT_DXB_NW4=1 timeout 200 python tools/prof_train.py 9000 80 2>&1 | grep "train step" >> $OUT/r06_p29_ab.txt
done
cat $OUT/r06_p29_ab.txt
unset RRT_HIP_LIB
bash tools/prof_train.sh r06_p29 9000 30 | grep -i "ln_backward\|bwd_dx\|train step"
