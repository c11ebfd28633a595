#!/bin/bash
# Round evidence: bench lines + rocprofv3 kernel stats (fp32: default 2-stream bench and 1-stream; bf16: 1-stream) + PMC
# passes (separate --pmc runs with --kernel-trace only, each under its own timeout).  Run on the GPU box from the repo root:
#     tools/profile_round.sh <tag>        -> gpurun_out/<tag>_*.{txt,json}
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
timeout 600 python $R/bench.py > $OUT/${TAG}_bench_default.bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
timeout 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_cmd.bench.json 2>/dev/null
for c in 2 3 4; do timeout 300 python $R/bench.py --config $c --no-cpu-baseline --steps 100 > $OUT/${TAG}_bench_config$c.bench.json 2>/dev/null; done
timeout 300 python $R/bench.py --dtype bf16 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_config1_bf16.bench.json 2>/dev/null
X="--no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python $R/bench.py $X > /tmp/a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_a/a_results.db > $OUT/${TAG}_bench_default_2streams.kernel_stats.txt
python $R/tools/rocprof_timeline.py /tmp/prof_a/a_results.db 48 < /dev/null | cut -c1-130 > $OUT/${TAG}_timeline_2streams.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $R/bench.py --streams 1 $X > /tmp/b.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_b/b_results.db > $OUT/${TAG}_bench_1stream.kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $R/bench.py --dtype bf16 --streams 1 $X > /tmp/c.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c/c_results.db > $OUT/${TAG}_bench_bf16_1stream.kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --config 3 --streams 1 --steps 50 $X > /tmp/d.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_d/d_results.db > $OUT/${TAG}_bench_config3_bf16_1stream.kernel_stats.txt
for dt in f32 bf16; do
  i=0
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/prof_${dt}_p$i -o p -- python $R/bench.py --dtype $dt --streams 1 --steps 5 --warmup 2 $X > /tmp/p$i.log 2>&1 || echo "pmc pass $dt $i ($pmc) failed/timeout"
    if [ -f /tmp/prof_${dt}_p$i/p_results.db ]; then
      echo "# pass: --dtype $dt --pmc $pmc" >> $OUT/${TAG}_pmc_$dt.txt
      python $R/tools/pmc_summary.py /tmp/prof_${dt}_p$i/p_results.db >> $OUT/${TAG}_pmc_$dt.txt
    fi
  done
done
timeout 300 python $R/bench.py --dtype f32x3 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_config1_f32x3.bench.json 2>/dev/null
timeout 300 python $R/bench.py --streams 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_1stream.bench.json 2>/dev/null
timeout 300 python $R/bench.py --streams 4 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_4streams.bench.json 2>/dev/null
if [ -f $R/tools/_abl/librrt_trace.so ]; then
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_fused.py > $OUT/${TAG}_trace_fused_f32_wave_timeline.txt 2>/dev/null
fi
[ -x $R/tools/_abl/mfma_valu_overlap ] && timeout 120 $R/tools/_abl/mfma_valu_overlap > $OUT/${TAG}_ubench_mfma_valu_overlap.txt 2>&1
tail -1 $OUT/${TAG}_bench_default.bench.json | cut -c1-200
head -14 $OUT/${TAG}_bench_1stream.kernel_stats.txt
