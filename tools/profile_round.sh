#!/bin/bash
# Round evidence: bench lines + rocprofv3 kernel stats (fp32: default 2-stream bench and 1-stream; bf16 / configs 3: 1-stream)
# + PMC passes (separate --pmc runs with --kernel-trace only, each under its own timeout) -> profiles-ready files.
# Run on the GPU box from the repo root:
#     tools/profile_round.sh <tag>        -> gpurun_out/<tag>_*.{txt,json} and gpurun_out/<tag>_traffic.json
# Order: the PMC passes first (they produce <tag>_traffic.json, which bench.py reads as profiles/<round>_traffic.json (round = the tag up to its first "_") when it
# is copied there BEFORE the bench lines are taken -- the script does that copy on the box so one call gives a consistent set).
set -u
TAG=${1:-r06_final}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_* /tmp/pmc_*
X="--no-cpu-baseline --no-extras"
rm -f $OUT/${TAG}_traffic.json
for spec in "c1_f32:--config 1 --dtype f32" "c1_bf16:--config 1 --dtype bf16" "c1_f32x3:--config 1 --dtype f32x3" \
            "c2_bf16:--config 2" "c3_bf16:--config 3" "c4_bf16:--config 4"; do
  key=${spec%%:*}; args=${spec#*:}
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pmc_${key}_$pmc -o p -- python $R/bench.py $args --streams 1 --steps 5 --warmup 2 $X > /tmp/pmc.log 2>&1 || echo "pmc pass $key $pmc failed/timeout"
  done
  if [ -f /tmp/pmc_${key}_FETCH_SIZE/p_results.db ] && [ -f /tmp/pmc_${key}_WRITE_SIZE/p_results.db ]; then
    python $R/tools/pmc_to_traffic.py $key /tmp/pmc_${key}_FETCH_SIZE/p_results.db /tmp/pmc_${key}_WRITE_SIZE/p_results.db $OUT/${TAG}_traffic.json "bench.py $args --streams 1 --steps 5 --warmup 2"
  fi
done
cp $OUT/${TAG}_traffic.json $R/profiles/${TAG%%_*}_traffic.json 2>/dev/null
# the dominant kernel's duration in a rocprofv3 kernel trace of the DEFAULT command (bench.py reads it back as roofline.rocprof)
rm -f $OUT/${TAG}_rocprof_dominant.json
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python $R/bench.py $X > /tmp/a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_a/a_results.db > $OUT/${TAG}_bench_default_4streams.kernel_stats.txt
python $R/tools/rocprof_timeline.py /tmp/prof_a/a_results.db 100 0.45 < /dev/null | cut -c1-250 > $OUT/${TAG}_timeline_4streams.txt
python $R/tools/rocprof_union.py c1_f32_s4 /tmp/prof_a/a_results.db $OUT/${TAG}_rocprof_dominant.json 22045261824 157.3
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_a2 -o a -- python $R/bench.py --streams 2 $X > /tmp/a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_a2/a_results.db > $OUT/${TAG}_bench_2streams.kernel_stats.txt
python $R/tools/rocprof_union.py c1_f32_s2 /tmp/prof_a2/a_results.db $OUT/${TAG}_rocprof_dominant.json 22045261824 157.3
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_a3 -o a -- python $R/bench.py --dtype bf16 $X > /tmp/a.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_a3/a_results.db > $OUT/${TAG}_bench_bf16_4streams.kernel_stats.txt
python $R/tools/rocprof_union.py c1_bf16_s4 /tmp/prof_a3/a_results.db $OUT/${TAG}_rocprof_dominant.json 17213423616 2500
cp $OUT/${TAG}_rocprof_dominant.json $R/profiles/${TAG%%_*}_rocprof_dominant.json 2>/dev/null
for dt in f32 bf16; do
  for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pmc_x
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pmc_x -o p -- python $R/bench.py --dtype $dt --streams 1 --steps 5 --warmup 2 $X > /tmp/pmc.log 2>&1 || echo "pmc pass $dt ($pmc) failed/timeout"
    if [ -f /tmp/pmc_x/p_results.db ]; then
      echo "# pass: --dtype $dt --pmc $pmc" >> $OUT/${TAG}_pmc_$dt.txt
      python $R/tools/pmc_summary.py /tmp/pmc_x/p_results.db >> $OUT/${TAG}_pmc_$dt.txt
    fi
  done
done
timeout 900 python $R/bench.py > $OUT/${TAG}_bench_default.bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
timeout 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_cmd.bench.json 2>/dev/null
for c in 2 3 4; do timeout 300 python $R/bench.py --config $c --no-cpu-baseline --steps 100 > $OUT/${TAG}_bench_config$c.bench.json 2>/dev/null; done
timeout 300 python $R/bench.py --dtype bf16 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_config1_bf16.bench.json 2>/dev/null
timeout 300 python $R/bench.py --dtype f32x3 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_config1_f32x3.bench.json 2>/dev/null
timeout 300 python $R/bench.py --streams 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_1stream.bench.json 2>/dev/null
timeout 300 python $R/bench.py --streams 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_2streams.bench.json 2>/dev/null
: > $OUT/${TAG}_streams_sweep.txt
for S in 1 2 3 4 5 6 8; do
  timeout 200 python $R/bench.py --streams $S $X 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 N=9000 streams=$S', r['value'], 'slides/s', r['ms_per_step'], 'ms/step', (r.get('value_spread') or {}).get('values'))" >> $OUT/${TAG}_streams_sweep.txt
done
for S in 2 3 4 5; do
  timeout 200 python $R/bench.py --dtype bf16 --streams $S $X 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 N=9000 streams=$S', r['value'], 'slides/s', r['ms_per_step'], 'ms/step', (r.get('value_spread') or {}).get('values'))" >> $OUT/${TAG}_streams_sweep.txt
done
timeout 400 python $R/tools/corun_matrix.py 40 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_corun_matrix.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $R/bench.py --streams 1 $X > /tmp/b.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_b/b_results.db > $OUT/${TAG}_bench_1stream.kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $R/bench.py --dtype bf16 --streams 1 $X > /tmp/c.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_c/c_results.db > $OUT/${TAG}_bench_bf16_1stream.kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --config 3 --streams 1 --steps 50 $X > /tmp/d.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_d/d_results.db > $OUT/${TAG}_bench_config3_bf16_1stream.kernel_stats.txt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o e -- python $R/bench.py --config 4 --streams 1 --steps 10 $X > /tmp/e.log 2>&1
python $R/tools/rocprof_summary.py /tmp/prof_e/e_results.db > $OUT/${TAG}_bench_config4_bf16_1stream.kernel_stats.txt
if [ -f $R/tools/_abl/librrt_trace.so ]; then
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_fused.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_fused_f32_wave_timeline.txt
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_fused_proj.py 9000 8 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_fused_proj_f32_wave_timeline.txt
  TRACE_STATS_K=3 RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_fused_proj.py 9000 8 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_fused_proj_stats_f32_wave_timeline.txt
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_combine_parts.py 9000 3 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_combine_parts_wave_timeline.txt
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_linear.py proj 9000 8 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_out_projection_f32_wave_timeline.txt
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_pair16.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_pair16_wave_timeline.txt
fi
[ -x $R/tools/_abl/dma_rows ] && timeout 120 $R/tools/_abl/dma_rows > $OUT/${TAG}_ubench_dma_rows.txt 2>&1
[ -x $R/tools/_abl/dma_loaders_mfma ] && timeout 120 $R/tools/_abl/dma_loaders_mfma > $OUT/${TAG}_ubench_dma_loaders_mfma.txt 2>&1
[ -x $R/tools/_abl/mfma_valu_overlap ] && timeout 120 $R/tools/_abl/mfma_valu_overlap > $OUT/${TAG}_ubench_mfma_valu_overlap.txt 2>&1
if [ -f $R/tools/_abl/librrt_trace.so ]; then
  RRT_HIP_LIB=$R/tools/_abl/librrt_trace.so timeout 120 python $R/tools/trace_pair16_proj.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_trace_pair16_proj_wave_timeline.txt
fi
bash $R/tools/prof_train.sh $TAG 9000 30 > /dev/null 2>&1
SWEEP_OUT=$OUT/${TAG}_sweep_n.txt; timeout 300 python $R/tools/sweep_n.py 2>/dev/null | grep -v amdgpu.ids > $SWEEP_OUT
SWEEP_DTYPE=bf16 timeout 300 python $R/tools/sweep_n.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_sweep_n_bf16.txt
tail -1 $OUT/${TAG}_bench_default.bench.json | cut -c1-300
head -14 $OUT/${TAG}_bench_1stream.kernel_stats.txt
