"""Per-iteration timeline of a rocprofv3 kernel trace (rocpd sqlite): for the LAST full iteration (between two consecutive
launches of the marker kernel) print every dispatch with start offset, duration and the idle gap before it."""
import sqlite3
import sys

db, marker = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_mf_score"
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t and "rocpd" in t][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
gcol = [x for x in cols if x in ("grid_size_x", "grid_x", "grid_size")]
gsel = ("d.%s" % gcol[0]) if gcol else "0"
qcol = [x for x in cols if x in ("queue_id", "stream_id")]
qsel = ("d.%s" % qcol[0]) if qcol else "0"
rows = list(c.execute("select s.kernel_name, d.start, d.end, %s, %s from %s d join %s s on d.kernel_id = s.id order by d.start" % (gsel, qsel, kd, sym)))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = marks[-2], marks[-1]
t0 = rows[a][1]
prev_end = rows[a][2]
busy = 0
print("iteration: %d dispatches, %.1f us wall" % (b - a, (rows[b][1] - t0) / 1e3))
agg = {}
for name, st, en, _g, _q in rows[a + 1:b + 1]:
    short = name.split("(")[0].replace("void ", "").replace("mfm::", "")[:40]
    gap = (st - prev_end) / 1e3
    agg.setdefault(short, [0, 0.0, 0.0])
    agg[short][0] += 1
    agg[short][1] += (en - st) / 1e3
    agg[short][2] += max(gap, 0.0)
    prev_end = max(prev_end, en)
print("%-42s %6s %10s %12s" % ("kernel", "calls", "busy_us", "gap_before_us"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-42s %6d %10.1f %12.1f" % (k, v[0], v[1], v[2]))
if "-v" in sys.argv:
    prev_end = rows[a][2]
    print("columns of the dispatch table:", cols)
    for name, st, en, g, q in rows[a + 1:b + 1]:
        print("%10.1f %8.1f gap %7.1f grid %8s q %4s  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, g, q, name.split("(")[0][-44:]))
        prev_end = max(prev_end, en)
