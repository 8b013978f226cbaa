#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- build container only (needs /root/reference).

SURVEY 8(d)-ii calibration: how much faster is the oracle's C restatement (what bench.py times on the GPU
box as `cpu_baseline`, kind "port") than the reference's own Cython module (utils/nms.pyx, which pays a
PyFloat allocation + rich compare per surviving pair, nms.pyx:65)?  The reference cannot travel to the GPU
box, so the ratio is measured HERE, on the same inputs, and committed as oracle/reference_ratio.json;
bench.py copies it into cpu_baseline.reference_ratio so that a reference-equivalent rate can be derived:
    reference-equivalent boxes/s  =  cpu_baseline.value / ratio.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def best_of(fn, reps):
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        d = time.perf_counter() - t
        best = d if best is None else min(best, d)
    return best


def main():
    import make_golden
    import synth
    from oracle import oracle
    R = make_golden.load_reference()
    oracle.build()
    rows = []
    for n, reps in ((300, 50), (2000, 10), (10000, 5)):
        d = synth.dets5(n, n)
        kr = R['nms'].nms(d, 0.3)
        ko = oracle.nms(d, 0.3)
        assert kr == ko
        t_ref = best_of(lambda: R['nms'].nms(d, 0.3), reps)
        t_port = best_of(lambda: oracle.nms(d, 0.3), reps)
        rows.append(dict(n=n, kept=len(kr), reference_ms=t_ref * 1e3, port_ms=t_port * 1e3, ratio=t_ref / t_port))
        print(rows[-1])
    out = dict(what="time(reference Cython utils/nms.pyx nms) / time(oracle/vdet_oracle.c oracle_nms), same inputs, thresh 0.3, "
                    "best of N, one core of the build container",
               host="build container: %d vCPU, %s" % (os.cpu_count(), open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t')),
               cases=rows, ratio_at_10k=rows[-1]['ratio'])
    with open(os.path.join(HERE, 'reference_ratio.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
