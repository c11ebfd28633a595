#!/bin/bash
# Kernel table of the encoder training step (rocprofv3 --kernel-trace --stats) -> gpurun_out/<tag>_train_kernel_stats.txt
# usage (GPU box): tools/prof_train.sh <tag> [N] [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-dev}; N=${2:-9000}; STEPS=${3:-30}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
# no pipes behind rocprofv3: a child that lingers at exit would keep them open past the timeout
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/tools/prof_train.py $N $STEPS > /tmp/prof_t.log 2>&1
echo "rocprofv3 rc=$?"; grep "train step" /tmp/prof_t.log
[ -f /tmp/prof_t/t_results.db ] && python $R/tools/rocprof_summary.py /tmp/prof_t/t_results.db > $OUT/${TAG}_train_kernel_stats.txt
head -40 $OUT/${TAG}_train_kernel_stats.txt < /dev/null | cut -c1-150
