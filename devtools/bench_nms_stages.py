"""Per-stage times of vdet_nms_volume alone on the bench video (graph build, sort, walk): python devtools/bench_nms_stages.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vdetlib_amd import ops
b, s = bench.synth_video_cuda(torch, 2000, 300, 10000, 200, "cuda")
ctx = ops._ctx_for(s)
for rep in range(3):
    ctx.set_timing(1)
    try:
        ops.nms_volume(b, s, 0.3)
    except Exception as e:
        print("raised", type(e).__name__, str(e)[:80])
    t = ctx.last_timing()
    print({k: round(v[0], 3) for k, v in t.items() if v[0] > 0}, flush=True)
