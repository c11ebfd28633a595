#!/bin/bash
# Round-end evidence: bench line + rocprofv3 kernel stats (default 2-stream bench and 1-stream) + PMC passes
# (separate --pmc runs with --kernel-trace only, each under its own timeout).  Run on the GPU box from the repo root:
#     tools/profile_round.sh <tag>        -> gpurun_out/<tag>_*.{txt,json}
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
timeout 600 python $R/bench.py > $OUT/${TAG}_bench_default.bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python $R/bench.py --no-cpu-baseline > /tmp/a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_a/a_results.db > $OUT/${TAG}_bench_default_2streams.kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $R/bench.py --streams 1 --no-cpu-baseline > /tmp/b.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_b/b_results.db > $OUT/${TAG}_bench_1stream.kernel_stats.txt
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/prof_p$i -o p -- python $R/bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/p$i.log 2>&1 || echo "pmc pass $i ($pmc) failed/timeout"
  if [ -f /tmp/prof_p$i/p_results.db ]; then
    echo "# pass: --pmc $pmc" >> $OUT/${TAG}_pmc.txt
    python $R/tools/pmc_summary.py /tmp/prof_p$i/p_results.db >> $OUT/${TAG}_pmc.txt
  fi
done
tail -1 $OUT/${TAG}_bench_default.bench.json | cut -c1-200
head -14 $OUT/${TAG}_bench_1stream.kernel_stats.txt
