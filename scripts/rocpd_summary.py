#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) --kernel-trace --stats result:
per-kernel calls / total / average / min / max duration, like the CSV stats table."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-60s %6s %14s %14s %12s %12s %6s %5s %5s %7s %9s %4s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "sgpr", "lds", "grid_x", "wg"))
for r in rows:
    print("%-60s %6d %14d %14.0f %12d %12d %6.2f %5d %5d %7d %9d %4d" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))
