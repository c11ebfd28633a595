#!/bin/bash
# tools/ab_proj.sh [tag]  -- on the GPU box: the out-projection as a phase of the fused launch (default) against the two
# launches (RRT_NO_FUSED_PROJ=1), tuning build, fp32 N = 9000, 1 / 2 / 4 bags in flight, interleaved twice
TAG=${1:-ab}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
X="--no-cpu-baseline --no-extras"
: > $OUT/${TAG}_proj.txt
for rep in 1 2; do
  for S in 1 2 4; do
    for mode in merged pair; do
      if [ $mode = pair ]; then export RRT_NO_FUSED_PROJ=1; else unset RRT_NO_FUSED_PROJ; fi
      timeout 200 python $R/bench.py --dtype f32 --streams $S $X > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
      python - <<PY >> $OUT/${TAG}_proj.txt
import json
try:
    r = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print("rep $rep S=$S $mode", r["value"], "slides/s  ms/step", r["ms_per_step"])
except Exception as e:
    print("rep $rep S=$S $mode: no bench line", e)
PY
    done
  done
done
cat $OUT/${TAG}_proj.txt
