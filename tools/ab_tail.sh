#!/bin/bash
# tools/ab_tail.sh <tag> [bench args] -- on the GPU box: A/B of tuning-build switches on the one-bag-in-flight kernel table.
#   VARIANTS="name:ENV=1,ENV2=x name2:" (default: the CR-MSA first pass from the slabs' records vs crmsa_region4)
# Each variant: rocprofv3 --kernel-trace --stats of `bench.py --streams 1` with the tuning library -> gpurun_out/<tag>_<name>.kernel_stats.txt
TAG=${1:-ab}; shift
ARGS=${@:-"--streams 1 --steps 40 --warmup 5"}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
VARIANTS=${VARIANTS:-"parts: region4:RRT_NO_CRMSA_PARTS=1"}
cd /tmp && export TMPDIR=/tmp
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
for v in $VARIANTS; do
  name=${v%%:*}; envs=${v#*:}
  rm -rf /tmp/prof_$name
  ( for e in ${envs//,/ }; do export $e; done
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py $ARGS --no-cpu-baseline --no-extras > /tmp/p_$name.json 2>/tmp/p_$name.err )
  python $R/tools/rocprof_summary.py /tmp/prof_$name/p_results.db > $OUT/${TAG}_${name}.kernel_stats.txt
  echo "== $name ($envs)"
  head -10 $OUT/${TAG}_${name}.kernel_stats.txt | cut -c1-60,96-150
  python - <<PY
import json
try:
    r = json.loads(open("/tmp/p_$name.json").read().strip().splitlines()[-1])
    print("   value", r["value"], "one bag", (r.get("one_bag_in_flight") or {}).get("ms_per_bag"))
except Exception as e:
    print("   no bench line", e)
PY
done
