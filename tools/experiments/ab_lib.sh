# tools/experiments/ab_lib.sh "<libA> <libB> ..." [bench args]  -- on the GPU box: bench.py lines (value + repeats) for
# tools/_abl/librrt_<lib>.so, twice round-robin, fp32 four bags in flight by default.
LIBS=$1; shift
ARGS=${@:-"--dtype f32 --steps 100 --warmup 10"}
R=$PWD
for rep in 1 2; do for lib in $LIBS; do
  RRT_HIP_LIB=$R/tools/_abl/librrt_$lib.so timeout 300 python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))"
done; done
