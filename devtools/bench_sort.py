#!/usr/bin/env python3
"""Times the per-(frame, class) sort alone (vdet_argsort_volume, class-major keys): python devtools/bench_sort.py [F] [B] [C] [kind]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vdetlib_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 200
kind = sys.argv[4] if len(sys.argv) > 4 else "rand"
g = torch.Generator(device="cuda").manual_seed(1)
s = (torch.rand if kind == "rand" else torch.randn)(F, C, B, generator=g, device="cuda")
ctx = ops._ctx_for(s)
for rep in range(3):
    ctx.set_timing(1)
    o, n = ops.argsort_volume(s, layout="FCB")
    t = ctx.last_timing()
    print(kind, (F, B, C), "sort %.3f ms  fallback kernel %.3f ms  columns handed over: %d" % (t["sort"][0], t["sort_fallback"][0], ctx.query(9)), flush=True)
ref = torch.argsort(s[:2], dim=2, descending=True, stable=True)     # (equal scores: the build orders by descending index)
print("same scores in the same order as torch.argsort on 2 frames:",
      bool((torch.gather(s[:2], 2, o[:2].to(torch.int64)) == torch.gather(s[:2], 2, ref)).all()))
