#!/usr/bin/env python3
"""Per (kernel name, grid size) durations of a rocprofv3 rocpd kernel trace - the same kernel at different stages of the tower (its
grid differs with the map size) shown separately.  usage: tools/rocpd_by_grid.py x_results.db [name-substring ...]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    gcols = [c for c in cols if "grid" in c.lower()]
    sel = ", ".join(gcols) if gcols else "0"
    rows = cur.execute(f"select {name_col}, {sel}, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                       f"group by {name_col}, {sel} order by 1, 2").fetchall()
    flt = sys.argv[2:]
    print(f"| kernel | grid ({', '.join(gcols)}) | calls | avg_us | min_us | max_us |\n|---|---|---:|---:|---:|---:|")
    for r in rows:
        n = r[0]
        if flt and not any(f in n for f in flt):
            continue
        ng = len(gcols) if gcols else 1
        g = " x ".join(str(v) for v in r[1:1 + ng])
        c, a, mn, mx = r[1 + ng:]
        print(f"| `{n[:70]}` | {g} | {c} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} |")


if __name__ == "__main__":
    main()
