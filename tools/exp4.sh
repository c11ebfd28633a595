#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp4.txt; : > $T
X="--no-cpu-baseline --no-extras"
python -m pytest tests -m gpu -x -q -k "rmsa_fused or cache or test_encoder_matches or concurrent or forward_bags or full_size or repeatable" 2>&1 | tail -8 | tee -a $T
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"], "frac", (r.get("roofline") or {}).get("frac"), "raw", (r.get("roofline") or {}).get("raw_interval_ms"), "iso", (r.get("roofline_isolated") or {}).get("avg_launch_ms"), "1bag", (r.get("one_bag_in_flight") or {}).get("ms_per_bag"), "spread", (r.get("value_spread") or {}).get("values"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for S in 1 2 3 4; do
  timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; line "fold S=$S" /tmp/b.json | tee -a $T
done
timeout 600 python bench.py --streams 4 > $OUT/exp4_full_S4.json 2>/tmp/b.err || tail -5 /tmp/b.err
line "full S=4" $OUT/exp4_full_S4.json | tee -a $T
