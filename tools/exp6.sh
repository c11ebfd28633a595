#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp6.txt; : > $T
X="--no-cpu-baseline --no-extras"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"], "frac", (r.get("roofline") or {}).get("frac"), "iso", (r.get("roofline_isolated") or {}).get("avg_launch_ms"), "1bag", (r.get("one_bag_in_flight") or {}).get("ms_per_bag"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
timeout 900 python bench.py > $OUT/exp6_default.json 2>/tmp/b.err || tail -5 /tmp/b.err
line "default" $OUT/exp6_default.json | tee -a $T
python - <<PY | tee -a $T
import json
r = json.loads(open("$OUT/exp6_default.json").read().strip().splitlines()[-1])
for k in ("module_call", "config0", "config2", "config3", "config4", "value_spread", "amp_bf16", "f32x3"):
    v = r.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("note", "workload", "stages")}
    print(k, json.dumps(v)[:600])
PY
for S in 2 3 4 5 6; do
  timeout 200 python bench.py --dtype bf16 --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "bf16 S=$S" /tmp/b.json | tee -a $T
done
for S in 3 4; do
  timeout 200 python bench.py --config 3 --streams $S --steps 60 $X > /tmp/b.json 2>/tmp/b.err; line "c3 S=$S" /tmp/b.json | tee -a $T
  timeout 200 python bench.py --config 4 --streams $S --steps 20 $X > /tmp/b.json 2>/tmp/b.err; line "c4 S=$S" /tmp/b.json | tee -a $T
  timeout 200 python bench.py --config 2 --streams $S --steps 100 $X > /tmp/b.json 2>/tmp/b.err; line "c2 S=$S" /tmp/b.json | tee -a $T
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee -a $T
