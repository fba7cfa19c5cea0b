"""The cube tool's bake on several GPUs of one node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m vulkan_renderer_amd.bake [--lambertian] [--ggx] [--cubemap-size S] [--input-downscale D] file.png

Same flags, defaults and output files as bin/cube (src/utilities/cube.cpp:392-517 in the
reference).  Every rank decodes the strip and bakes its shard of every output cube; the
ranks pack their shards' texels, one all-gather per cube moves them and a scatter puts them
in place; rank 0 encodes to RGBE and writes X.lambertian.png / X.ggx<m>.png.
A texel's value does not depend on the number of ranks, so the files hold the same
pixels as a single-GPU run of bin/cube.  Run without a launcher it is a one-GPU bake.
"""
import argparse
import os
import sys


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--lambertian", action="store_true")
    ap.add_argument("--ggx", action="store_true")
    ap.add_argument("--cubemap-size", type=int, default=16)
    ap.add_argument("--input-downscale", type=int, default=1)
    ap.add_argument("file")
    args = ap.parse_args(argv)
    d = args.input_downscale
    if d < 1 or (d & (d - 1)):
        ap.error("scale factor for --input-downscale should be a power of 2 >= 1")   # cube.cpp:433-435
    return args


def main(argv=None):
    args = parse_args(argv)
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import capi, pngio, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: every rank on cuda:0 over gloo (RCCL refuses two ranks per device)
    backend = os.environ.get("IBL_BAKE_BACKEND", "nccl")
    if os.environ.get("IBL_BAKE_SAME_DEVICE") == "1":
        local_rank = 0
    # IBL_BAKE_EXCHANGE_ALWAYS=1: a single rank also joins a process group and runs pack, all-gather and
    # scatter (one rank over RCCL is what a one-GPU box can run of the real exchange step)
    always = os.environ.get("IBL_BAKE_EXCHANGE_ALWAYS") == "1"
    torch.cuda.set_device(local_rank)
    grouped = world > 1 or always
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    ibl = capi.Ibl(local_rank)

    def assemble(kind, in_w, cube, mip=1, mips=1):
        # over gloo on one shared GPU (rehearsal) the shards travel through host tensors: see sharding.gather_cube
        return sharding.gather_cube(dist, ibl, kind, in_w, cube, mip, mips, rank, world, via_host=backend == "gloo",
                                    always=always)

    strip = pngio.read_rgba8(args.file)
    h6, w = strip.shape[0], strip.shape[1]
    if w == 0 or h6 != 6 * w:
        raise SystemExit("input must be a W x 6W strip of square cube faces")
    env = ibl.rgbe_decode(torch.from_numpy(strip).to(ibl.device)).reshape(6, w, w, 4).contiguous()
    if args.input_downscale > 1:
        env = ibl.downscale(env, args.input_downscale)                      # cube.cpp:472-504
    stem = os.path.splitext(args.file)[0]

    def write(name, cube, size):
        if rank == 0:
            rgbe = ibl.rgbe_encode(cube).cpu().numpy().reshape(6 * size, size, 4)
            pngio.write_rgba8(stem + name, rgbe)
            print("wrote %s%s" % (os.path.basename(stem), name), flush=True)

    if args.lambertian:                                                      # cube.cpp:509-511
        size = args.cubemap_size
        out = torch.zeros((6, size, size, 4), dtype=torch.float32, device=ibl.device)
        ibl.lambertian_shard(env, size, rank, world, out=out)
        write(".lambertian.png", assemble(capi.SHARD_LAMBERTIAN, int(env.shape[2]), out), size)
    if args.ggx:                                                             # cube.cpp:512-514
        mips = ibl.ggx_mip_levels(w, w, args.cubemap_size)                   # from the pre-downscale size, :473-474
        for m in range(1, mips + 1):
            size = w >> m
            out = torch.zeros((6, size, size, 4), dtype=torch.float32, device=ibl.device)
            ibl.ggx_level_shard(env, size, m, mips, rank, world, out=out)
            write(".ggx%d.png" % m, assemble(capi.SHARD_GGX, int(env.shape[2]), out, m, mips), size)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
