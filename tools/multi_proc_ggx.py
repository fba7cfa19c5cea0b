"""Every rank bakes its shard of a GGX chain on cuda:0 at the same time, with and
without a collective after each level, under a watchdog (rehearsal of the multi-GPU
path on a one-GPU box; launch with torch.distributed.run)."""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import vulkan_renderer_amd as vra

faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG", "50")), exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("BACKEND", "gloo"), rank=rank, world_size=world)
ibl = vra.Ibl(0)
# letters: t = timing on, n = no synchronize before the collective, c = counting pass first,
# d = a plain device-to-host copy instead of the collective, m = a torch matmul chain instead of the GGX kernel,
# s = the kernel and the collective on a side stream instead of the default stream
MODE = os.environ.get("MODE", "")
if "t" in MODE:
    ibl.set_timing(True)
in_w = 512
strip = vra.synth.hdr_environment(in_w, seed=1234)
env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
mips = ibl.ggx_mip_levels(in_w, in_w, 16)
if "c" in MODE:
    ibl.set_counting(True)
    for m in range(1, mips + 1):
        ibl.ggx_level(env, in_w >> m, m, mips)
        ibl.last_counts()
    ibl.set_counting(False)
    print("rank %d counting pass done" % rank, flush=True)
side = torch.cuda.Stream() if "s" in MODE else None
if side is not None:
    torch.cuda.synchronize()
    torch.cuda.set_stream(side)
big = torch.randn(4096, 4096, device="cuda") if "m" in MODE else None
for collective in (False, True):
    for m in range(1, mips + 1):
        ow = in_w >> m
        out = torch.zeros((6, ow, ow, 4), device="cuda")
        if big is not None:
            acc = big
            for _ in range(30):
                acc = (acc @ big) * 1e-2
            out += acc[0, 0]
        else:
            ibl.ggx_level_shard(env, ow, m, mips, rank, world, out=out)
        if "t" in MODE and m == 1:
            ibl.last_kernel_ms()
        if "n" not in MODE:
            torch.cuda.synchronize()
        print("rank %d level %d kernel done (collective=%s)" % (rank, m, collective), flush=True)
        if collective and "d" in MODE:
            host = out.cpu()
            print("rank %d level %d device-to-host copy done (%g)" % (rank, m, float(host.sum())), flush=True)
        elif collective:
            dist.all_reduce(out)
            torch.cuda.synchronize()
            print("rank %d level %d all_reduce done" % (rank, m), flush=True)
dist.barrier()
dist.destroy_process_group()
