#!/usr/bin/env python3
"""Per-stage times of the batched VID-shape run (and of one video alone): python devtools/bench_vid.py [V]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bench
from vdetlib_amd import ops, _lib
V = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
boxes, scores, off = bench.synth_vid_batch(torch, dev, V)
kw = dict(nms_thres=0.3, thres=0.5, max_tracks=4, link_thres=0.5)
cx = _lib.Context(0); cx.set_cache(True)
TAPS = [0.25, 0.5, 0.25]
for rep in range(2):
    cx.invalidate(); cx.set_timing(2)
    t0 = time.perf_counter()
    p, c = ops.volume_pass(scores, 3, TAPS, ctx=cx, frame_off=off)
    out = ops.video_batch(boxes, scores, off, cap=300, overlap_thres=0.7, window=3, ctx=cx, pad=False, **kw)
    torch.cuda.synchronize()
    print("batch of %d videos (%d frames): wall %.1f ms" % (V, off[-1], (time.perf_counter() - t0) * 1e3))
    print({k: (round(ms, 3), n) for k, (ms, n) in cx.last_timing().items() if n})
    cx.set_timing(0)
vb, vs = boxes[:off[1]], scores[:off[1]]
for rep in range(2):
    cx.invalidate(); cx.set_timing(2)
    t0 = time.perf_counter()
    p, c = ops.volume_pass(vs, 3, TAPS, ctx=cx)
    ki, kc, tr, an, nt = ops.nms_track_volume(vb, vs, cap=300, ctx=cx, pad=False, **kw)
    det, tp, tb = ops.rescore_tracks(tr, nt, vb, vs, overlap_thres=0.7, window=3, ctx=cx)
    torch.cuda.synchronize()
    print("one video (%d frames): wall %.2f ms" % (off[1], (time.perf_counter() - t0) * 1e3))
    print({k: (round(ms, 3), n) for k, (ms, n) in cx.last_timing().items() if n})
    cx.set_timing(0)
print("ntracks", nt.tolist()[:8], "track length", int((~torch.isnan(tr[0, 0, :, 0])).sum()))
