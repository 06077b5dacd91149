"""Summarise a rocprofv3 rocpd database (kernel-trace) into a text table: python scripts/prof_summary.py db out.txt"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
    f.write("# command: %s\n" % (sys.argv[3] if len(sys.argv) > 3 else ""))
    f.write("%-100s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        f.write("%-100s %8d %14.3f %12.3f %8.3f\n" % (name[:100], calls, tot, avg, pct))
print(open(out).read()[:3000])
