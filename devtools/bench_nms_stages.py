"""Per-stage times of vdet_nms_volume alone on the bench video (graph build, sort, walk): python devtools/bench_nms_stages.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vdetlib_amd import ops
b, s = bench.synth_video_cuda(torch, 2000, 300, 10000, 200, "cuda")
ctx = ops._ctx_for(s)
track = len(sys.argv) > 1 and sys.argv[1] == "track"      # the lists with the exact heads the tracking kernels read
for rep in range(3):
    ctx.set_timing(1)
    try:
        if track:
            ops.nms_track_volume(b, s, nms_thres=0.3, thres=0.9, max_tracks=10, cap=2048, pad=False, ctx=ctx)
        else:
            ops.nms_volume(b, s, 0.3)
    except Exception as e:
        print("raised", type(e).__name__, str(e)[:80])
    t = ctx.last_timing()
    print({k: round(v[0], 3) for k, v in t.items() if v[0] > 0}, flush=True)
