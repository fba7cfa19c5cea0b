"""all_reduce of CUDA tensors of growing size between ranks that share cuda:0 (gloo),
under a watchdog: separates a collective problem from a kernel problem in rehearsals."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(40, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("IBL_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
for mb in (1, 2, 4, 8, 16, 32):
    t = torch.full((mb * 262144,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert t[0].item() == world * (world + 1) / 2
    if rank == 0:
        print("all_reduce %d MB ok" % mb, flush=True)
dist.destroy_process_group()
