#!/bin/bash
# round 6, probe 21: rmsa_pair16<9, ., 2> at 110 VGPRs (U fragments in batches of three; -DRRT_PAIR16_LEAN_A=3) against the product's 119:
# does a co-resident wave of another bag's streaming kernel (35-56 VGPRs) pay for the shorter MFMA runs?  bf16, four bags in flight + kernel time alone
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
RRT_HIP_LIB=$R/tools/_abl/librrt_lean3.so timeout 600 python -m pytest tests -m gpu -x -q -k "fused16 or autocast or two_bags" 2>&1 | tail -2
: > $OUT/r06_p21_ab.txt
bash tools/experiments/ab_lib.sh "base lean3" --dtype bf16 --steps 30 --warmup 5 2>&1 | sed 's/^/bf16 /' >> $OUT/r06_p21_ab.txt
bash tools/experiments/ab_lib.sh "base lean3" --config 2 --steps 100 --warmup 5 2>&1 | sed 's/^/c2 /' >> $OUT/r06_p21_ab.txt
cat $OUT/r06_p21_ab.txt
cd /tmp; export TMPDIR=/tmp
for lib in base lean3; do
  rm -rf /tmp/prof_x; RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype bf16 --streams 1 --steps 40 --no-cpu-baseline --no-extras > /tmp/p.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "rmsa_pair16" | cut -c1-60,96-140 | sed "s/^/$lib /"
done
