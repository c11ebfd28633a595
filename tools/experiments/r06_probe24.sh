#!/bin/bash
# round 6, probe 24: the non-temporal hint on the LDS-DMA reads of the 16-bit intermediates on their LAST read (u16 into rmsa_pair16's
# qkv projection, O16 into the out-projection): same-box A/B of four builds, bags in flight (bf16 N = 9000, configs 2-4)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { lbl=$1; lib=$2; a=$3; ( [ -n "$lib" ] && export RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so
    timeout 300 python bench.py $a --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lbl', r['value'], r['value_spread']['values'])" ); }
: > $OUT/r06_p24_ab.txt
for rep in 1 2 3; do
  for lib in "" ntproj ntpair ntboth; do
    run "bf16 ${lib:-base}" "$lib" "--dtype bf16 --steps 30 --warmup 5" >> $OUT/r06_p24_ab.txt
  done
  for lib in "" ntproj ntpair ntboth; do
    run "c3 ${lib:-base}" "$lib" "--config 3 --steps 12 --warmup 3" >> $OUT/r06_p24_ab.txt
  done
  for lib in "" ntboth; do
    run "c2 ${lib:-base}" "$lib" "--config 2 --steps 100 --warmup 5" >> $OUT/r06_p24_ab.txt
    run "c4 ${lib:-base}" "$lib" "--config 4 --steps 30 --warmup 5" >> $OUT/r06_p24_ab.txt
  done
done
cat $OUT/r06_p24_ab.txt
