"""scratch: sort cost on quantised (tie-heavy) scores vs continuous ones"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from vdetlib_amd import ops, _lib
F, B, C = 60, 10000, 200
g = torch.Generator(device='cuda').manual_seed(3)
x1 = torch.rand(F, B, generator=g, device='cuda') * 1230; y1 = torch.rand(F, B, generator=g, device='cuda') * 670
w = 10 + torch.rand(F, B, generator=g, device='cuda') * 290; h = 10 + torch.rand(F, B, generator=g, device='cuda') * 290
boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=1279), torch.clamp(y1 + h, max=719)], -1).round().contiguous()
cx = _lib.get_context(0)
for name, sc in (('continuous', torch.rand(F, B, C, generator=g, device='cuda')),
                 ('quantised 1/64', (torch.rand(F, B, C, generator=g, device='cuda') * 64).round() / 64)):
    for rep in range(3):
        cx.set_timing(1)
        ops.nms_volume(boxes, sc, 0.3, cap=10000)
        t = cx.last_timing()
    print(name, 'sort ms', round(t['sort'][0], 3), 'walk ms', round(t['walk'][0], 3))
