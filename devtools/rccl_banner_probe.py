import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from vdetlib_amd import dist as vdist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
vdist.init(backend="nccl", device=dev, force=True)
t = torch.ones(4, device=dev); out = torch.empty(4, device=dev)
dist.all_gather_into_tensor(out, t); torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group()
print("PROBE_DONE", os.environ.get("RCCL_LOG_LEVEL"))
