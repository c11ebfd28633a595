#!/bin/bash
# round 6, first probe on the GPU box: (1) fresh vs cycled inputs through the raw C-ABI loop, (2) forward_bags by caller stream /
# outputs, (3) the all-no-packed-fp32 build, (4) the training step's kernel table.   -> gpurun_out/r06_probe1.txt
R=$PWD; OUT=$R/gpurun_out/r06_probe1.txt; : > $OUT
X="--no-cpu-baseline --no-extras"
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('one_bag_in_flight') or {}).get('ms_per_bag'), r['config'].get('fresh_inputs'))" >> $OUT; }
for dt in f32 bf16; do
  timeout 200 python bench.py --dtype $dt $X 2>/dev/null | line "raw $dt cycled"
  timeout 200 python bench.py --dtype $dt --fresh-inputs $X 2>/dev/null | line "raw $dt fresh"
done
timeout 200 python bench.py --config 3 --steps 40 $X 2>/dev/null | line "raw c3 cycled"
timeout 200 python bench.py --config 3 --steps 40 --fresh-inputs $X 2>/dev/null | line "raw c3 fresh"
timeout 200 python bench.py --config 2 --steps 60 $X 2>/dev/null | line "raw c2 cycled"
timeout 200 python bench.py --config 2 --steps 60 --fresh-inputs $X 2>/dev/null | line "raw c2 fresh"
for dt in f32 bf16; do
  for spec in "256 cyc fresh default" "256 cyc alias default" "256 cyc fresh side" "256 cyc alias side" "256 fresh fresh side" "64 cyc fresh default" "64 cyc fresh side" "64 fresh fresh side"; do
    timeout 120 python tools/experiments/r06_probe_bags.py $dt $spec 2>/dev/null >> $OUT
  done
done
for rep in 1 2; do for lib in hip nopk; do
  L=$R/tools/_abl/librrt_$lib.so; [ $lib = hip ] && L=$R/rrt-mil_amd/librrt_hip.so
  for dt in f32 bf16; do
    RRT_HIP_LIB=$L timeout 200 python bench.py --dtype $dt --steps 100 --warmup 10 $X 2>/dev/null | line "lib=$lib $dt"
  done
done; done
tools/prof_train.sh r06_base 9000 30 >> $OUT 2>&1
cat $OUT | cut -c1-220
