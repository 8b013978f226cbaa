#!/usr/bin/env python3
"""Timing experiment: the LSD radix sort (LDS-pipe bound) and the counting sort (VALU bound) of DISJOINT halves of one
volume's columns in flight together on two streams -- do their workgroups share CUs to any profit?
    python devtools/bench_sort_mix.py [F] [B] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vdetlib_amd import _lib


def make_ctx(binsort):
    os.environ["VDET_BINSORT"] = "1" if binsort else "0"
    return _lib.Context(torch.cuda.current_device())


F = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 200
g = torch.Generator(device="cuda").manual_seed(1)
s = torch.rand(F, C, B, generator=g, device="cuda")
order = torch.empty((F, C, B), dtype=torch.int16, device="cuda")
ncand = torch.zeros((F, C), dtype=torch.int32, device="cuda")
ctxs = {0: make_ctx(False), 1: make_ctx(True), 2: make_ctx(False), 3: make_ctx(True)}
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def launch(ctx, st, f0, f1):
    if f1 <= f0:
        return
    ctx.set_stream(st.cuda_stream)
    ctx.check(ctx.lib.vdet_argsort_volume(ctx.h, s[f0:f1].data_ptr(), _lib.LAYOUT_FCB, f1 - f0, B, C, 0, 0.0,
                                          order[f0:f1].data_ptr(), ncand[f0:f1].data_ptr()))


def run(kinds, split):
    """kinds: (ctx id on stream 0, ctx id on stream 1); split: frames on stream 0"""
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        launch(ctxs[kinds[0]], streams[0], 0, split)
        launch(ctxs[kinds[1]], streams[1], split, F)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1))
    return best


print("all LSD, one stream      %.3f ms" % run((0, 2), F), flush=True)
print("all counting, one stream %.3f ms" % run((1, 3), F), flush=True)
print("LSD + LSD halves         %.3f ms" % run((0, 2), F // 2), flush=True)
print("counting + counting      %.3f ms" % run((1, 3), F // 2), flush=True)
for frac in (0.3, 0.4, 0.5, 0.6, 0.7):
    print("LSD %.0f %% + counting %.0f %%  %.3f ms" % (100 * frac, 100 - 100 * frac, run((0, 1), int(F * frac))), flush=True)
ref = torch.argsort(s[:1], dim=2, descending=True, stable=True)     # (equal scores: the build orders by descending index)
print("same scores in the same order as torch.argsort on frame 0:",
      bool((torch.gather(s[:1], 2, order[:1].to(torch.int64)) == torch.gather(s[:1], 2, ref)).all()))
