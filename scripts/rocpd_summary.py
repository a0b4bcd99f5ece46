"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as text."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select * from top_kernels"))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (view top_kernels of %s)\n" % db.split("/")[-1])
    if len(sys.argv) > 3:
        f.write("# command: %s\n" % sys.argv[3])
    f.write("%-10s %12s %12s %7s  %s\n" % ("calls", "total_us", "avg_us", "pct", "kernel"))
    for name, calls, total, avg, pct in rows:
        f.write("%-10d %12.1f %12.3f %7.2f  %s\n" % (calls, total, avg, pct, name))
print(open(out).read()[:1500])
