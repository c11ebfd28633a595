#!/bin/bash
# round 6, probe 10: crmsa_stream4 for EVERY region size (librrt_s4all: -DRRT_STREAM4_MIN_P=16) against the product's rule (region4 up to 144 tokens)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
RRT_HIP_LIB=$R/tools/_abl/librrt_s4all.so timeout 900 python -m pytest tests -m gpu -x -q -k "golden or forward_bags or config4 or autocast or oracle_f64" 2>&1 | tail -3 > $OUT/r06_p10_tests.txt; cat $OUT/r06_p10_tests.txt
: > $OUT/r06_p10_ab.txt
bash tools/experiments/ab_lib.sh "base s4all" --dtype bf16 --steps 30 --warmup 5 2>&1 | sed 's/^/bf16 /' >> $OUT/r06_p10_ab.txt
bash tools/experiments/ab_lib.sh "base s4all" --config 4 --steps 30 --warmup 5 2>&1 | sed 's/^/c4 /' >> $OUT/r06_p10_ab.txt
bash tools/experiments/ab_lib.sh "base s4all" --config 2 --steps 100 --warmup 5 2>&1 | sed 's/^/c2 /' >> $OUT/r06_p10_ab.txt
bash tools/experiments/ab_lib.sh "base s4all" --dtype f32 --steps 10 --warmup 3 2>&1 | sed 's/^/f32 /' >> $OUT/r06_p10_ab.txt
cat $OUT/r06_p10_ab.txt
cd /tmp; export TMPDIR=/tmp
for lib in base s4all; do
  rm -rf /tmp/prof_x; RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype bf16 --streams 1 --steps 40 --no-cpu-baseline --no-extras > /tmp/p.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "crmsa_" | cut -c1-60,96-140 | sed "s/^/$lib /"
done
