"""Timing probe of the volume pass alone: how much does the placement of its four streams (1 read, 3 writes of 2.4 GB)
matter?  python devtools/vpass_probe.py"""
import sys, time, torch
sys.path.insert(0, '.')
from vdetlib_amd import ops, _lib
dev = torch.device('cuda', 0)
F, B, C = 300, 10000, 200
cx = _lib.Context(0)
cx.set_cache(True)
pad = []
for trial in range(8):
    pad.append(torch.empty(1 << (12 + trial), dtype=torch.uint8, device=dev))   # shift the allocator's next addresses
    s = torch.rand(F, B, C, device=dev)
    for _ in range(2):
        ops.volume_pass(s, 3, [0.25, 0.5, 0.25], ctx=cx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        p, c = ops.volume_pass(s, 3, [0.25, 0.5, 0.25], ctx=cx)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        del p, c
    print('trial', trial, 'scores@%x' % (s.data_ptr() & 0xFFFFFF), 'ms', ['%.2f' % t for t in ts])
    del s
