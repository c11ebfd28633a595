# A/B on the GPU box: crmsa_dispatch_ln_kernel with one token per wave (tune) against two tokens per wave sharing the
# representatives' rows (rw2), one bag in flight, fp32 + bf16; parity of both first.
R=$PWD
for lib in tune rw2; do
  RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so python -m pytest $R/tests/test_hip_parity.py -q -m gpu -x -k "dispatch or encoder_forward" 2>&1 | tail -1
done
cd /tmp; export TMPDIR=/tmp
for lib in tune rw2 tune rw2; do
  for dt in f32 bf16; do
    rm -rf /tmp/prof_x
    RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype $dt --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > /tmp/p.json 2>/dev/null
    python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "dispatch_ln" | cut -c1-45,96-125 | sed "s/^/$lib $dt  /"
  done
done
