#!/bin/bash
# tools/sweep_streams.sh [tag] [dtype]  -- on the GPU box: slides/s over bags in flight (streams), with and without the phase gate
TAG=${1:-sw}
DT=${2:-f32}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
X="--no-cpu-baseline --no-extras"
: > $OUT/${TAG}_streams_sweep.txt
for S in 1 2 3 4 5 6 8; do
  for gate in 0 1; do
    if [ $gate = 1 ] && [ $S = 1 ]; then continue; fi
    RRT_BENCH_GATE=$gate timeout 200 python $R/bench.py --dtype $DT --streams $S $X > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
    python - <<PY >> $OUT/${TAG}_streams_sweep.txt
import json
try:
    r = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print("$DT S=$S gate=$gate", r["value"], "slides/s  ms/step", r["ms_per_step"])
except Exception as e:
    print("$DT S=$S gate=$gate: no bench line", e)
PY
  done
done
cat $OUT/${TAG}_streams_sweep.txt
