#!/usr/bin/env python3
"""Per-kernel SQ counters from one rocprofv3 PMC pass (8 SQ slots per pass, MI355X_MICROARCH.md):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_sq -o s -- python bench.py ...
    python profiles/sq_summarize.py gpurun_out/pmc_sq/s_results.db profiles/r02_pmc_sq.csv
Values are means per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of the resident waves' time spent issuing VALU),
wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier)."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    acc = {}
    names = set()
    for kn, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if "vdet::" not in kn:
            continue
        short = kn.split("(")[0].replace("void ", "")
        a = acc.setdefault(short, {})
        c = a.setdefault(cn, [0, 0.0])
        c[0] += 1
        c[1] += val
        names.add(cn)
    names = sorted(names)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc %s ; means per launch\n" % " ".join(names))
        f.write("kernel,launches," + ",".join(names) + ",valu_issue_frac,wait_frac\n")
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [1, 0])[1]):
            n = max(c[0] for c in a.values())
            mean = {cn: (a[cn][1] / a[cn][0] if cn in a and a[cn][0] else 0.0) for cn in names}
            wc = mean.get("SQ_WAVE_CYCLES", 0.0)
            f.write('"%s",%d,%s,%.3f,%.3f\n' % (k, n, ",".join("%.4g" % mean[cn] for cn in names),
                                                 mean.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else 0.0,
                                                 mean.get("SQ_WAIT_ANY", 0.0) / wc if wc else 0.0))
    print(open(out).read())


if __name__ == "__main__":
    main()
