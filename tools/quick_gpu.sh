#!/bin/bash
# tools/quick_gpu.sh <tag> [pytest -k expr]  -- on the GPU box: selected GPU tests, then one-bag-in-flight kernel stats
# (rocprofv3) and bench lines for fp32 / bf16 / configs 3, 4 into gpurun_out/<tag>_*
TAG=${1:-q}
KEXPR=${2:-}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
if [ -n "$KEXPR" ]; then
  python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -12 > $OUT/${TAG}_tests.txt
else
  python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/${TAG}_tests.txt
fi
cat $OUT/${TAG}_tests.txt
cd /tmp && export TMPDIR=/tmp
X="--no-cpu-baseline --no-extras"
for spec in "f32:--dtype f32" "bf16:--dtype bf16" "c3:--config 3 --steps 60" "c4:--config 4 --steps 30" "c2:--config 2 --steps 100"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 300 python $R/bench.py $args $X > $OUT/${TAG}_bench_$name.json 2>/tmp/err_$name.log || tail -3 /tmp/err_$name.log
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/${TAG}_bench_$name.json").read().strip().splitlines()[-1])
    print("$name", r["value"], r["unit"], "ms/step", r["ms_per_step"], "frac", (r.get("roofline") or {}).get("frac"))
except Exception as e:
    print("$name: no bench line", e)
PY
done
for spec in "f32:--dtype f32 --streams 1" "bf16:--dtype bf16 --streams 1" "c3:--config 3 --streams 1 --steps 40"; do
  name=${spec%%:*}; args=${spec#*:}
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py $args $X > /tmp/p_$name.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/prof_$name/p_results.db > $OUT/${TAG}_${name}_1stream.kernel_stats.txt
  head -12 $OUT/${TAG}_${name}_1stream.kernel_stats.txt | cut -c1-150
done
