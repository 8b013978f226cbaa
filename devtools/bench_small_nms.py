"""scratch: latency of the host-buffer drop-in (cython_nms.nms) at small N"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import synth
from vdetlib_amd.utils import cython_nms as K
for n in (100, 300, 1000, 2000, 10000):
    d = synth.dets5(n, n)
    K.nms(d, 0.3)
    t = time.perf_counter()
    reps = 50 if n <= 2000 else 10
    for _ in range(reps):
        k = K.nms(d, 0.3)
    dt = (time.perf_counter() - t) / reps
    print('n=%5d  %.1f us per call  kept %d' % (n, dt * 1e6, len(k)))
