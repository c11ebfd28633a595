#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp2.txt; : > $T
X="--no-cpu-baseline --no-extras"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"], "spread", (r.get("value_spread") or {}).get("values"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for S in 2 3 4 5 6 8; do
  timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "head S=$S" /tmp/b.json | tee -a $T
done
for S in 3 4 6 8; do
  RRT_BENCH_SOLO=1 timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "head solo=1 S=$S" /tmp/b.json | tee -a $T
done
for S in 4 6; do
  RRT_BENCH_GATE=1 timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "head gate S=$S" /tmp/b.json | tee -a $T
done
