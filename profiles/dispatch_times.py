#!/usr/bin/env python3
"""Per-dispatch durations of the kernels matching a substring, in launch order, from a rocprofv3 --kernel-trace
sqlite result (rocpd):  python profiles/dispatch_times.py results.db track_link [max_rows]"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = None
for t in tabs:
    if 'kernel' in t.lower():
        cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
        if 'start' in cols and 'end' in cols and any(c in cols for c in ('name', 'kernel_name')):
            nm = 'name' if 'name' in cols else 'kernel_name'
            rows = list(cur.execute("select %s, start, end from %s order by start" % (nm, t)))
            break
if rows is None:
    print("tables:", tabs)
    sys.exit(1)
sel = [(n, s, e) for n, s, e in rows if pat in n]
print("# %d dispatches of *%s* (us), in launch order" % (len(sel), pat))
print(" ".join("%.0f" % ((e - s) / 1e3) for n, s, e in sel[:lim]))
