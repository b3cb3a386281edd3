#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean / share, by name+grid.
usage: rocpd_stats.py results.db [--by-grid] > profiles/xxx_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
by_grid = "--by-grid" in sys.argv
key = "name, grid_x, grid_y, grid_z" if by_grid else "name"
rows = db.execute(f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                  f"group by {key} order by sum(duration) desc").fetchall()
tot = sum(r[-4] for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[-5] for r in rows)} dispatches")
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'share':>6}  kernel")
for r in rows:
    name = r[0]
    n, s, a, mn, mx = r[-5:]
    short = name if len(name) < 150 else name[:147] + "..."
    grid = f" grid=({r[1]},{r[2]},{r[3]})" if by_grid else ""
    print(f"{n:7d} {s / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:5.1f}%  {short}{grid}")
