#!/usr/bin/env python3
"""Per-kernel PMC counter means from a rocprofv3 rocpd sqlite db.  usage: rocpd_pmc.py results.db [name-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
rows = db.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
agg = {}
for r in rows:
    name = r[ci.get('kernel_name', ci.get('name', 0))]
    if sub and sub not in str(name): continue
    key = (str(name)[:90], r[ci['counter_name']])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r[ci['value']])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:90s} {c:28s} n={n:3d} mean={v / n:16.1f}")
