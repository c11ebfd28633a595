#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp8.txt; : > $T
python -m pytest tests -m gpu -x -q -k "batch_gt_1 or input_ranks or attn_pool or rrtmil or pool_predict or attention64 or crmsa_stages or golden" 2>&1 | tail -6 | tee -a $T
for NB in 64 256 1024; do for S in 2 4; do
  echo -n "bags nb=$NB S=$S: " | tee -a $T
  GPU_MAX_HW_QUEUES=16 timeout 300 python tools/bench_bags.py uniform $NB $S 2>/dev/null | tail -1 | tee -a $T
done; done
timeout 900 python bench.py > $OUT/exp8_default.json 2>/tmp/b.err || tail -5 /tmp/b.err
python - <<PY | tee -a $T
import json
r = json.loads(open("$OUT/exp8_default.json").read().strip().splitlines()[-1])
print("default", r["value"], r["ms_per_step"], r["roofline"]["frac"], r.get("value_spread", {}).get("values"))
for k in ("module_call", "config0", "config2", "config3", "config4", "rrtmil_c16", "train_step"):
    v = r.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("note", "workload", "stages", "roofline", "roofline_isolated", "command")}
    print(k, json.dumps(v)[:500])
PY
timeout 300 python tools/bench_train.py 2>/dev/null | tail -8 | tee -a $T
