#!/usr/bin/env python3
"""Print the per-kernel summary (calls, total/avg us, %) of a rocprofv3 rocpd SQLite result."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
print(f"# rocprofv3 --kernel-trace --stats summary of {db.split('/')[-1]} (durations in microseconds)")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, tot, avg, pct in rows:
    print(f"{calls:6d} {tot:12.3f} {avg:12.3f} {pct:7.3f}  {name}")
