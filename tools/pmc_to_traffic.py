#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two rocpd .db files) -> one record of profiles/<round>_traffic.json.

    python tools/pmc_to_traffic.py <key> <fetch.db> <write.db> <out.json> [note]

Per kernel: dispatches, mean FETCH_SIZE and WRITE_SIZE (KiB, as rocprofv3 reports them) and
hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE in bytes -- FETCH_SIZE doubled per the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (128-byte requests tallied at 64 B).  bench.py reads the file: `roofline.traffic`
and `roofline_kernels[*].traffic` are looked up here by (config, dtype), never typed in by hand."""
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection").fetchall()
    per = defaultdict(float)
    dur = {}
    for k, d, c, v, du in rows:
        if c == counter:
            per[(k, d)] += v
            dur[(k, d)] = du
    agg = defaultdict(list)
    durs = defaultdict(list)
    for (k, d), v in per.items():
        agg[k].append(v)
        durs[k].append(dur[(k, d)])
    return {k: (sum(v) / len(v), len(v), sum(durs[k]) / len(durs[k])) for k, v in agg.items()}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    key, fdb, wdb, out = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if k.startswith(("at::", "__amd_rocclr")):
            continue
        fk, n, du = f.get(k, (0.0, 0, 0.0))
        wk, n2, du2 = w.get(k, (0.0, 0, 0.0))
        kernels[short(k)] = {"dispatches": max(n, n2), "fetch_size_kib": round(fk, 1), "write_size_kib": round(wk, 1),
                             "hbm_bytes": int(round((2.0 * fk + wk) * 1024)),
                             "duration_us_in_counter_pass": round(max(du, du2) / 1e3, 2)}
    data = {}
    if os.path.exists(out):
        with open(out) as fh:
            data = json.load(fh)
    data[key] = {"note": note, "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE; per "
                 "dispatch means; hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction of the guide)", "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)
    print(f"{key}: {len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
