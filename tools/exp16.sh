#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp16.txt; : > $T
export RRT_NO_CRMSA_REGION4=1
for L in guard guardn; do
  for M in f32x3 bf16 f32; do
    RRT_HIP_LIB=$R/tools/_abl/librrt_$L.so REPRO_MODE=$M timeout 300 python tools/repro_guarded_ln.py 60 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $T
  done
done
unset RRT_NO_CRMSA_REGION4
