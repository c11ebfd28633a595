R=$PWD; cd /tmp; export TMPDIR=/tmp
for lib in tune pkcr tune pkcr; do
  for dt in f32 bf16; do
    rm -rf /tmp/prof_x
    RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype $dt --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > /tmp/p.json 2>/dev/null
    python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | grep "crmsa_\|ln_partition" | cut -c1-45,96-125 | sed "s/^/$lib $dt  /"
  done
done
