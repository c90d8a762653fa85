#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into a `--stats`-style table (per kernel: calls,
total / average / min / max duration, share) and, if present, per-kernel PMC counter sums.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# kernel-trace summary of `{path.split('/')[-1]}` (durations in microseconds)\n")
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx in rows:
        n = n if len(n) < 110 else n[:107] + "..."
        print(f"| `{n}` | {c} | {s/1e3:.1f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/total:.2f} |")
    try:   # one row per (dispatch, counter, dimension instance): sum the instances, then aggregate over dispatches
        pm = cur.execute("select name, counter_name, count(*), sum(v) from (select name, counter_name, dispatch_id, "
                         "sum(counter_value) as v from pmc_events group by name, counter_name, dispatch_id) "
                         "group by 1,2 order by 1,2").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n## PMC counters (sum over dispatches)\n\n| kernel | counter | dispatches | sum | per dispatch |\n|---|---|---:|---:|---:|")
        for n, cn, c, s in pm:
            n = n if len(n) < 90 else n[:87] + "..."
            print(f"| `{n}` | {cn} | {c} | {s:.6g} | {s/c:.6g} |")


if __name__ == "__main__":
    main(sys.argv[1])
