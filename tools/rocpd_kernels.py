#!/usr/bin/env python3
"""tools/rocpd_kernels.py <results.db> [marker-substring] -- kernels of a rocprofv3 run (its SQLite output) in launch
order with start offsets and durations: the last sequence that ends with a kernel whose name holds the marker."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
marker = sys.argv[2] if len(sys.argv) > 2 else "stream_boundaries"
idx = [i for i, r in enumerate(rows) if marker in r[0]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 1
lo = idx[which - 1] + 1 if which > 0 else 0
hi = idx[which + 1] if which + 1 < len(idx) else len(rows)
t0 = rows[lo][1]
for n, s, e, g, w in rows[lo:hi]:
    print("  +%9.1f us %9.1f us  grid %8d wg %4d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, g, w, n.split("(")[0]))
