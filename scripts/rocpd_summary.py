#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, %) of a rocprofv3 rocpd SQLite database - the same
aggregation `rocprofv3 --stats` prints as kernel_stats.csv.  usage: rocpd_summary.py results.db [steps] > summary.md"""
import sqlite3
import sys

db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for name, n, t, a, mn, mx in rows:
    print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f |" % (name[:120], n, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
print()
print("total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)) +
      (" = %.3f ms per step (%d steps)" % (tot / 1e6 / steps, steps) if steps else ""))
