#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 --kernel-trace sqlite result: for the steady-state part of a
multi-stream run, how long NO kernel ran, how long exactly one ran, and which kernels ran alone.
    python profiles/timeline.py results.db [skip_fraction [first_step last_step]]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
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
rows = [(n.split('(')[0].replace('void ', '').replace('vdet::', ''), s, e) for n, s, e in rows]
t0, t1 = rows[0][1], max(e for _, _, e in rows)
lo, hi = t0 + (t1 - t0) * skip, t1
if len(sys.argv) > 4:      # window = from the a-th to the b-th dispatch of the volume pass (one per step): the timed steps
    a, b = int(sys.argv[3]), int(sys.argv[4])
    vp = [s for n, s, e in rows if n.startswith('volume_pass')]
    lo, hi = vp[a], vp[b]
ev = []
for n, s, e in rows:
    if e <= lo or s >= hi:
        continue
    s = max(s, lo); e = min(e, hi)
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
active = defaultdict(int)
nact = 0
last = lo
hist = defaultdict(float)
alone = defaultdict(float)
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        hist[min(nact, 4)] += dt
        if nact == 1:
            k = [a for a, c in active.items() if c > 0][0]
            alone[k] += dt
    last = t
    active[n] += d
    nact += d
tot = sum(hist.values())
print("window %.1f ms" % (tot / 1e6))
for k in sorted(hist):
    print("  %d kernel(s) running%s: %.1f%%" % (k, "+" if k == 4 else "", 100 * hist[k] / tot))
print("running alone (share of the window):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-50s %.1f%%" % (k[:50], 100 * v / tot))
