"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) per kernel.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db [--skip-first N] > profiles/xyz.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        st = stats.setdefault(name, [])
        st.append((e - s) / 1e3)
    total = sum(sum(v) for v in stats.values())
    print(f"# {path}: {len(rows)} kernel dispatches, {total / 1e3:.3f} ms total GPU kernel time")
    print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'pct':>6s}")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{short(name):92s} {len(v):6d} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f} "
              f"{sum(v) / 1e3:9.3f} {100 * sum(v) / total:6.2f}")


if __name__ == "__main__":
    main()
