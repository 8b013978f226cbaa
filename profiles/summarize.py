#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel summary committed under profiles/.
    python profiles/summarize.py gpurun_out/prof_r1/r1_results.db profiles/r01_kernel_stats.csv "command line"
"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats ; %s\n" % cmd)
    f.write("# durations in microseconds\n")
    f.write("kernel,calls,total_us,avg_us,percent\n")
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 90:
            short = short[:87] + "..."
        f.write('"%s",%d,%.1f,%.1f,%.2f\n' % (short, calls, tot, avg, pct))
print(open(out).read())
