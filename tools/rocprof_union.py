"""The dominant kernel's duration in a rocprofv3 --kernel-trace db of a bench.py run, two ways:
mean dispatch duration, and the union of the dispatches' [start, end] intervals divided by their number (what bench.py's
`roofline.avg_launch_ms` measures with HIP events when several bags are in flight and launches of different streams
overlap).  Merges one record into a json file keyed like bench.py looks it up.

    python tools/rocprof_union.py <key e.g. c1_f32_s4> <results.db> <out.json> <flops_per_launch> <peak_tflops> [pattern ...]
Only the middle 40 % .. 90 % of the run's dispatches of that kernel are used (warm-up and the one-bag-in-flight passes at the end
of a bench run are not the timed region)."""
import json
import os
import sqlite3
import sys


def main():
    key, dbp, outp, flops, peak = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), float(sys.argv[5])
    pats = sys.argv[6:] or ["rmsa_fused_kernel", "rmsa_pair16", "rmsa_fused16_kernel<9", "rmsa_fused16_kernel<8", "rmsa_fused_x3"]
    db = sqlite3.connect(dbp)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(n, s, e) for n, s, e in rows if any(p in n for p in pats)]
    if not rows:
        print("no dispatch matches", pats)
        return
    n = len(rows)
    win = rows[int(n * 0.40):int(n * 0.90)] if n >= 40 else rows
    name = max(set(r[0] for r in win), key=lambda k: sum(1 for r in win if r[0] == k))
    win = [r for r in win if r[0] == name]
    mean_ms = sum(e - s for _, s, e in win) / len(win) / 1e6
    iv = sorted((s, e) for _, s, e in win)
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    union_ms = busy / len(win) / 1e6
    rec = {"kernel": name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), "dispatches": len(win),
           "mean_dispatch_ms": round(mean_ms, 5), "avg_launch_ms": round(union_ms, 5),
           "frac": round(flops / (union_ms * 1e-3) / 1e12 / peak, 4),
           "frac_of_mean_dispatch": round(flops / (mean_ms * 1e-3) / 1e12 / peak, 4),
           "command_db": os.path.basename(os.path.dirname(dbp)) or os.path.basename(dbp)}
    out = {}
    if os.path.exists(outp):
        with open(outp) as fh:
            out = json.load(fh)
    out[key] = rec
    with open(outp, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(key, json.dumps(rec))


if __name__ == "__main__":
    main()
