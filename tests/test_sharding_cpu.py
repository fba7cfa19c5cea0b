"""CPU tests of the multi-GPU host logic: the partitions and the one all-gather per cube
that assembles a sharded bake, run as two gloo ranks.  The per-rank compute is
stood in for by the oracle (the HIP path needs a GPU); what is under test is the
partition/gather code that bench.py runs between the HIP calls."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from vulkan_renderer_amd import sharding  # noqa: E402


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 96, 98304, 98305):
        for world in (1, 2, 3, 8):
            seen = np.zeros(n, np.int32)
            prev_end = 0
            for r in range(world):
                b, c = sharding.shard_range(n, r, world)
                assert b == prev_end or c == 0
                seen[b:b + c] += 1
                prev_end = b + c if c else prev_end
            assert (seen == 1).all()


def test_shard_partitions_cover_every_texel_once():
    """The library's own partitions (host arithmetic of libibl_hip.so, no GPU needed): strided units of 64
    orbits of 48 texels (or contiguous octet ranges, by the kernel that serves the sizes) for Lambertian,
    strided units of 64 hexadecets or octets for GGX, plain ranges where the general kernels serve the sizes."""
    from vulkan_renderer_amd import capi
    for kind, in_w, out_w, mip, mips in ((capi.SHARD_LAMBERTIAN, 64, 32, 1, 1), (capi.SHARD_GGX, 64, 32, 1, 2),
                                         (capi.SHARD_GGX, 512, 256, 1, 5), (capi.SHARD_LAMBERTIAN, 30, 5, 1, 1),
                                         (capi.SHARD_GGX, 100, 50, 1, 2), (capi.SHARD_LAMBERTIAN, 1024, 128, 1, 1),
                                         (capi.SHARD_GGX, 2048, 16, 7, 7)):
        for n in (1, 2, 3, 8):
            per_shard = [capi.shard_indices(kind, in_w, out_w, mip, mips, r, n) for r in range(n)]
            assert len({len(a) for a in per_shard}) == 1          # equal packed sizes: one all-gather
            cat = np.concatenate(per_shard)
            valid = cat[cat != 0xffffffff]
            assert len(valid) == 6 * out_w * out_w and len(np.unique(valid)) == len(valid)
            # mirror images stay together wherever the octet kernels serve the sizes
            if in_w % 2 == 0 and out_w % 2 == 0 and (out_w & (out_w - 1)) == 0 and (in_w & (in_w - 1)) == 0:
                for a in per_shard:
                    v = a[a != 0xffffffff]
                    assert len(v) % 8 == 0
    lib = capi.load_library()
    assert lib.ibl_shard_texels(7, 64, 32, 1, 1, 2) == 0 and b"kind" in lib.ibl_last_error()
    assert lib.ibl_shard_texels(capi.SHARD_GGX, 8, 4, 1, 1, 2) == 0
    assert lib.ibl_shard_indices(capi.SHARD_GGX, 64, 32, 1, 2, 2, 2, None) != 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_units_list, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import vulkan_renderer_amd as vra
    from helpers import decode_strip
    env = decode_strip(oracle, vra.synth.hdr_environment(16, seed=9))
    ok = True
    # Lambertian 16 -> 5x5/face: 150 texels, an odd split
    for out_w in n_units_list:
        n_out = 6 * out_w * out_w
        b, c = sharding.shard_range(n_out, rank, world)
        mine = oracle.lambertian(env, out_w, out_begin=b, out_count=c)
        full = torch.from_numpy(mine.reshape(-1, 4).copy())
        sharding.all_gather_ranges(dist, full, n_out, rank, world)
        want = oracle.lambertian(env, out_w)
        ok = ok and np.array_equal(full.numpy(), want.reshape(-1, 4))
        # GGX level through the same gather
        mine = oracle.ggx_level(env, out_w, out_w, 1, 1, out_begin=b, out_count=c)
        full = torch.from_numpy(mine.reshape(-1, 4).copy())
        sharding.all_gather_ranges(dist, full, n_out, rank, world)
        ok = ok and np.array_equal(full.numpy(), oracle.ggx_level(env, out_w, out_w, 1, 1).reshape(-1, 4))
    # the collective the Lambertian / GGX product path uses: the library's shard of the cube
    # packed, one all-gather, every shard scattered into place
    from vulkan_renderer_amd import capi
    for kind, out_w, make in ((capi.SHARD_LAMBERTIAN, 8, lambda: oracle.lambertian(env, 8)),
                              (capi.SHARD_GGX, 8, lambda: oracle.ggx_level(env, 8, 8, 1, 1))):
        want = torch.from_numpy(make().reshape(-1, 4).copy())
        idx = [capi.shard_indices(kind, 16, out_w, 1, 1, r, world) for r in range(world)]
        mine = torch.zeros_like(want)                       # this rank holds only its own shard's texels
        own = idx[rank][idx[rank] != 0xffffffff].astype(np.int64)
        mine[own] = want[own]
        packed = sharding.pack_by_indices(mine, idx[rank])
        every = sharding.all_gather_packed(dist, packed, world, always=True)      # (always: a lone rank gathers too)
        assert every.shape[0] == world * len(idx[rank])
        full = sharding.unpack_by_indices(every, np.concatenate(idx), torch.full_like(want, -1.0))
        ok = ok and bool(torch.equal(full, want))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(int(flag.item()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_bake_equals_single_rank():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, [5, 8], ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


@pytest.mark.timeout(300)
def test_one_rank_runs_the_exchange_when_asked():
    """always=True: a single rank packs, all-gathers (a group of one) and scatters, which is how a
    one-GPU box runs the RCCL exchange; without it a single rank returns its buffers untouched."""
    packed = torch.arange(12.0).reshape(3, 4)
    assert sharding.all_gather_packed(None, packed, 1) is packed
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), [5], ret))
    p.start()
    p.join(240)
    assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it: the parent spawns torch.distributed.run
    (one rank per GPU, rendezvous on 127.0.0.1) and passes its arguments on; a process that already is
    a rank, or a one-GPU run, launches nothing."""
    import bench
    argv = ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    cmd, env = bench.self_launch_command(8, argv, {"PATH": "/usr/bin"}, port=23456)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "23456"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin" and "WORLD_SIZE" not in env
    assert bench.self_launch_command(1, ["--gpus", "1"], {}) is None
    assert bench.self_launch_command(8, argv, {"WORLD_SIZE": "8", "RANK": "3"}) is None
    cmd2, _ = bench.self_launch_command(2, ["--gpus", "2"], {})
    assert 1024 < int(cmd2[cmd2.index("--master-port") + 1]) < 65536


@pytest.mark.timeout(120)
def test_bench_passes_a_failed_rank_on():
    """The parent's exit status is the launcher's: a rank that fails (here: no GPU) makes
    `python bench.py --gpus 2` fail, and nothing is printed on stdout."""
    import subprocess
    env = dict(os.environ, IBL_BENCH_BACKEND="gloo", IBL_BENCH_SAME_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extras", "--no-cpu-baseline", "--workload", "c1"],
                       env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
