#!/bin/bash
# GPU experiment 1 (round 4): baselines at S = 1..4, co-run cost matrix, persistent fused blocks, region4 shapes
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; T=$OUT/exp1.txt; : > $T
X="--no-cpu-baseline --no-extras"
python -m pytest tests -m gpu -x -q -k "test_rmsa_fused or test_encoder_matches_reference_golden or cache_follows or test_forward_bags_mixed" 2>&1 | tail -5 | tee -a $T
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "ms/step", r["ms_per_step"], "frac", (r.get("roofline") or {}).get("frac"), "iso", (r.get("roofline_isolated") or {}).get("frac"), "1bag", (r.get("one_bag_in_flight") or {}).get("ms_per_bag"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for S in 1 2 3 4; do
  timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "prod S=$S" /tmp/b.json | tee -a $T
done
echo "--- corun matrix (product lib)" | tee -a $T
timeout 300 python tools/corun_matrix.py 40 2>&1 | grep -v amdgpu.ids | tee -a $T
export RRT_HIP_LIB=$R/tools/_abl/librrt_tune.so
for P in 256 248; do
for S in 1 2 3 4; do
  RRT_FUSED_PERSIST=$P timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "persist$P S=$S" /tmp/b.json | tee -a $T
done; done
for S in 2 4; do
  RRT_FUSED_PERSIST=256 RRT_REGION4_CFG=8 timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "persist256+r4cfg8 S=$S" /tmp/b.json | tee -a $T
  RRT_REGION4_CFG=8 timeout 200 python bench.py --streams $S $X > /tmp/b.json 2>/tmp/b.err; line "r4cfg8 S=$S" /tmp/b.json | tee -a $T
done
echo "--- corun matrix (persistent 256)" | tee -a $T
RRT_FUSED_PERSIST=256 timeout 300 python tools/corun_matrix.py 40 2>&1 | grep -v amdgpu.ids | tee -a $T
unset RRT_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_a
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python $R/bench.py --streams 2 --steps 60 $X > /tmp/a.log 2>&1
python - <<'PY' | tee -a $T
import sqlite3
db = sqlite3.connect("/tmp/prof_a/a_results.db")
print([r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")][:40])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
rows = db.execute("select * from kernels order by start limit 3").fetchall()
for r in rows: print(r)
PY
python $R/tools/rocprof_timeline.py /tmp/prof_a/a_results.db 60 0.3 | cut -c1-200 | tee -a $T
