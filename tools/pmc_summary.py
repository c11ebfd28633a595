"""Per-kernel mean of rocprofv3 --pmc counters (rocpd sqlite .db).

    python tools/pmc_summary.py gpurun_out/pmc1/r_results.db [kernel-substring]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection").fetchall()
    per = defaultdict(lambda: defaultdict(float))      # (kernel, dispatch) -> counter -> summed value
    dur = {}
    for k, d, c, v, du in rows:
        per[(k, d)][c] += v
        dur[(k, d)] = du
    agg = defaultdict(lambda: defaultdict(list))
    for (k, d), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
        agg[k]["_duration_ns"].append(dur[(k, d)])
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1]["_duration_ns"])):
        if filt not in k:
            continue
        name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:80]
        print(f"{name}  (n={len(cs['_duration_ns'])})")
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
