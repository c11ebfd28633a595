#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp5.txt; : > $T
X="--no-cpu-baseline --no-extras"
python -m pytest tests -m gpu -x -q -k "rmsa_fused_ln or cache or test_encoder_matches_reference" 2>&1 | tail -3 | tee -a $T
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"], "frac", (r.get("roofline") or {}).get("frac"), "iso", (r.get("roofline_isolated") or {}).get("avg_launch_ms"), "1bag", (r.get("one_bag_in_flight") or {}).get("ms_per_bag"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for L in prod nofold nostats; do
  unset RRT_HIP_LIB RRT_NO_LNFOLD
  if [ $L = nofold ]; then export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so RRT_NO_LNFOLD=1; fi
  if [ $L = nostats ]; then export RRT_HIP_LIB=$R/tools/_abl/librrt_nostats.so; fi
  for S in 1 2 4; do
    timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; line "$L S=$S" /tmp/b.json | tee -a $T
  done
done
