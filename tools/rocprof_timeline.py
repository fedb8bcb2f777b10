"""Dump a window of the kernel timeline of a rocprofv3 --kernel-trace run (rocpd sqlite): start/end in us, queue, name.

usage: python tools/rocprof_timeline.py <dir-or-db> [first_index] [count]      (TIMELINE_SKIP=<regex>: kernels left out, e.g. fills and copies)
"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = list(con.execute("select name, queue_id, start, end, grid_x from kernels order by start"))
if os.environ.get("TIMELINE_SKIP"):
    import re

    rows = [r for r in rows if not re.search(os.environ["TIMELINE_SKIP"], r[0])]
t0 = rows[first][2]
for name, q, s, e, gx in rows[first:first + count]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} us  q{q} grid={gx:<8} {name[:70]}")
