# A/B on the GPU box: librrt_tune.so (product flags) against librrt_kpl.so (-mllvm -amdgpu-kernarg-preload-count=16),
# one bag in flight, fp32 and bf16: kernel table + one_bag_in_flight.
R=$PWD; cd /tmp; export TMPDIR=/tmp
for lib in tune kpl tune kpl; do
  for dt in f32 bf16; do
    rm -rf /tmp/prof_x
    RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 python $R/bench.py --dtype $dt --streams 1 --steps 60 --warmup 10 --no-cpu-baseline --no-extras > /tmp/p.json 2>/dev/null
    python - <<PY
import json
r = json.loads(open("/tmp/p.json").read().strip().splitlines()[-1])
print("$lib $dt", r["value"], (r.get("one_bag_in_flight") or {}).get("ms_per_bag"))
PY
  done
done
for lib in tune kpl; do
  rm -rf /tmp/prof_x
  RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $R/bench.py --dtype f32 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > /tmp/p.json 2>/dev/null
  python $R/tools/rocprof_summary.py /tmp/prof_x/p_results.db | head -9 | cut -c1-45,96-125 | sed "s/^/$lib  /"
done
