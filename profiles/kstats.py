#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) per kernel: count / avg / min / max / total (us).
   kstats.py <db> [rows] [instances]: with `instances` (how many object instances the traced run processed) a column of
   device microseconds per instance is added -- a batch driver's per-instance cost decomposed by kernel."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0, "
                 "sum(end-start)/1000.0, max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                 "from kernels group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
per = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
head = "%-78s %5s %10s %10s %10s %11s %6s %5s %7s %9s %5s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct", "vgpr", "lds", "grid", "wg")
print(head + ("  us/instance" if per else ""))
for r in rows[:n_rows]:
    line = "%-78s %5d %10.2f %10.2f %10.2f %11.1f %6.2f %5d %7d %9d %5d" % (r[0][:78], r[1], r[2], r[3], r[4], r[5], 100 * r[5] / tot, r[6], r[7], r[8], r[9])
    print(line + ("  %11.2f" % (r[5] / per) if per else ""))
if per:
    print("%-78s %5s %10s %10s %10s %11.1f %6s %5s %7s %9s %5s  %11.2f" % ("all kernels (device time, summed over streams)", "", "", "", "", tot, "", "", "", "", "", tot / per))
