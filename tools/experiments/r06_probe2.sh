#!/bin/bash
# round 6, second probe: GPU suite on the no-packed build, the new bench line (forward_bags timed region), packed-fp32 reproducer
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/r06_p2_tests.txt; cat $OUT/r06_p2_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_p2_bench_driver.json 2> $OUT/r06_p2_bench_driver.err; tail -3 $OUT/r06_p2_bench_driver.err
X="--no-cpu-baseline --no-extras"
for spec in "f32:--dtype f32" "f32fresh:--dtype f32 --fresh-inputs" "f32raw:--dtype f32 --raw-loop" "bf16:--dtype bf16" "bf16fresh:--dtype bf16 --fresh-inputs" "bf16raw:--dtype bf16 --raw-loop" "c3:--config 3 --steps 20" "c4:--config 4 --steps 30" "c0:--config 0 --steps 100"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 300 python bench.py $args $X 2>/tmp/err_$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', r['value'], r['ms_per_step'], (r.get('value_spread') or {}).get('values'), (r.get('raw_c_abi_loop') or {}).get('value'), (r.get('roofline') or {}).get('frac'))" >> $OUT/r06_p2_lines.txt 2>&1 || tail -3 /tmp/err_$name.log >> $OUT/r06_p2_lines.txt
done
cat $OUT/r06_p2_lines.txt
tools/repro_packed_fp32.sh 1000 40
