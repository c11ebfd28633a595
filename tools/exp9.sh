#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp9.txt; : > $T
python -m pytest tests -m gpu -x -q -k "attn_pool" 2>&1 | grep -E "^E|passed|failed" | head -20 | tee -a $T
export GPU_MAX_HW_QUEUES=16
for A in "" 4 8; do
  echo -n "bags nb=256 S=4 alias='$A': " | tee -a $T
  BAGS_ALIAS_OUTS=$A timeout 300 python tools/bench_bags.py uniform 256 4 2>/dev/null | tail -1 | tee -a $T
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o x -- python $R/tools/bench_bags.py uniform 256 4 > /tmp/x.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_x/x_results.db | head -14 | cut -c1-150 | tee -a $T
python $R/tools/rocprof_timeline.py /tmp/prof_x/x_results.db 90 0.5 | cut -c1-260 > $OUT/exp9_timeline_exec_S4.txt
