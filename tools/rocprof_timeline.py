"""Steady-state timeline of a multi-stream bench run from a rocprofv3 --kernel-trace db: which kernels overlap.

    python tools/rocprof_timeline.py /tmp/prof/x_results.db [n_rows] [skip_fraction]
Prints, for a window in the middle of the run, every dispatch with start / end relative to the window and its queue.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name.split("(")[0][:44]


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select {namecol}, start, end, {qcol if qcol else 0} from kernels order by start"
    rows = db.execute(sel).fetchall()
    i0 = int(len(rows) * frac)
    # start the window at a fused kernel
    while i0 < len(rows) and "rmsa_fused" not in rows[i0][0]:
        i0 += 1
    t0 = rows[i0][1]
    qs = {}
    print(f"# columns: {cols}")
    for name, s, e, q in rows[i0:i0 + n]:
        qi = qs.setdefault(q, len(qs))
        pad = " " * (58 * qi)
        print(f"{pad}{short(name):44s} {(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f}")


if __name__ == "__main__":
    main()
