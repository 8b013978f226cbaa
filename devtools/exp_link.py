import sys, torch, json
sys.path.insert(0,'.')
from bench import synth_video_cuda
from vdetlib_amd import ops, _lib
dev=torch.device('cuda',0)
boxes,scores=synth_video_cuda(torch,2000,300,10000,200,dev)
ctx=_lib.get_context(0); ctx.set_cache(True)
for mf in (0,100,20,2):
    ctx.invalidate()
    ops.track_volume(boxes,scores,thres=0.9,max_tracks=10,max_frames=mf)
    ctx.set_timing(2)
    ctx.invalidate()
    tr,an,nt=ops.track_volume(boxes,scores,thres=0.9,max_tracks=10,max_frames=mf)
    t=ctx.last_timing(); ctx.set_timing(0)
    import numpy as np
    lens=(~torch.isnan(tr[:,:,:,0])).sum(-1).float().mean().item()
    print(mf, 'avg track len', round(lens,1), {k:(round(v[0],2),v[1]) for k,v in t.items() if v[1] and k.startswith('track')})
