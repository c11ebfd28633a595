# tools/experiments/ab_env.sh "<ENV=1> <ENV2=1> ..." [bench args] -- on the GPU box: bench.py lines of the tuning build with and
# without each switch (round-robin, twice).  "-" = no switch.
SW=$1; shift
ARGS=${@:-"--dtype f32 --steps 100 --warmup 10"}
R=$PWD
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
for rep in 1 2; do for sw in $SW; do
  ( [ "$sw" != "-" ] && export $sw; timeout 300 python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw', r['value'], r['value_spread']['values'], (r.get('one_bag_in_flight') or {}).get('ms_per_bag'))" )
done; done
