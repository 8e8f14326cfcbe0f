#!/usr/bin/env python3
"""tools/launch_series.py results.db substr [substr...] -- per-launch duration series (in launch order) of the kernels whose
name contains one of the substrings, from a rocprofv3 rocpd database: `t_ms  duration_us  grid  name`."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
t0 = rows[0][1]
for n, s, d, g in rows:
    short = n.split("(")[0].replace("void ", "").replace("mgpt::fastk::", "")[:48]
    if any(k in n for k in sys.argv[2:]):
        print("%10.3f %9.1f %9d %s" % ((s - t0) / 1e6, d / 1e3, g, short))
