#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp3.txt; : > $T
X="--no-cpu-baseline --no-extras"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"], "iso", (r.get("roofline_isolated") or {}).get("avg_launch_ms"), "1bag", (r.get("one_bag_in_flight") or {}).get("ms_per_bag"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for L in prod nt st3 nt3; do
  if [ $L = prod ]; then unset RRT_HIP_LIB; else export RRT_HIP_LIB=$R/tools/_abl/librrt_$L.so; fi
  if [ $L = st3 ]; then python -m pytest tests -m gpu -x -q -k "test_rmsa_fused or test_encoder_matches_reference_golden" 2>&1 | tail -2 | tee -a $T; fi
  for S in 1 2 4; do
    timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "$L S=$S" /tmp/b.json | tee -a $T
  done
  echo "--- corun $L" | tee -a $T
  timeout 300 python tools/corun_matrix.py 40 2>&1 | grep -v amdgpu.ids | grep -E "alone|ln_part|region4|dispatch|tail|rep_attn" | tee -a $T
done
unset RRT_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_a
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python $R/bench.py --streams 4 --steps 60 $X > /tmp/a.log 2>&1
python $R/tools/rocprof_timeline.py /tmp/prof_a/a_results.db 120 0.4 | cut -c1-260 > $OUT/exp3_timeline_S4.txt
python $R/tools/rocprof_summary.py /tmp/prof_a/a_results.db | head -20 | cut -c1-160 | tee -a $T
