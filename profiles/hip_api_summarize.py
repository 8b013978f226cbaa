#!/usr/bin/env python3
"""HIP runtime API call counts / time from a rocprofv3 --hip-runtime-trace sqlite result:
    python profiles/hip_api_summarize.py results.db out.csv
What it is for: showing that the asynchronous video step (vdet_set_async) issues no hipStreamSynchronize /
hipDeviceSynchronize / blocking copy between the entry and the return of the volume entry points -- the only
synchronising calls left in a bench run are the fences around the timed region and vdet_sync."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = None
for t in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
    if 'name' in cols and 'start' in cols and 'end' in cols and ('region' in t.lower() or 'api' in t.lower()):
        try:
            rows = list(cur.execute("select name, count(*), sum(end - start) from %s group by name order by 3 desc" % t))
        except sqlite3.Error:
            rows = None
        if rows:
            break
if not rows:
    print("no API table found; tables:", tabs)
    sys.exit(1)
with open(out, "w") as f:
    f.write("# rocprofv3 --hip-runtime-trace ; python bench.py --steps 8 --warmup 4 --no-cpu (3 streams, asynchronous builds)\n")
    f.write("api,calls,total_us\n")
    for n, c, t in rows:
        f.write("%s,%d,%.1f\n" % (n, c, (t or 0) / 1e3))
print(open(out).read())
