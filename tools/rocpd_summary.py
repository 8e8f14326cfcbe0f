#!/usr/bin/env python3
"""tools/rocpd_summary.py results.db [out.txt] -- per-kernel stats table from a rocprofv3 rocpd database
(the `--kernel-trace --stats` summary in text form: calls, total/avg/min/max duration, share, registers)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                     max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x)
                     from kernels group by name order by sum(duration) desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = [f"{'kernel':86s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} "
         f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>9s} {'wg':>5s}"]
for r in rows:
    lines.append(f"{r[0][:86]:86s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:10.2f} {r[5] / 1e3:10.2f} "
                 f"{100 * r[2] / tot:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:9d} {r[11]:5d}")
txt = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
print(txt)
