"""CPU tests of the multi-GPU host logic: range partition and the one all-gather
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


def test_weighted_ranges_balance_cost():
    # GGX levels: cost per output row falls with the level; ranges must be contiguous and balanced
    costs = [8.0] * 100 + [1.0] * 400 + [0.25] * 800
    for world in (1, 2, 4, 8):
        ranges = sharding.weighted_shard_ranges(costs, world)
        assert len(ranges) == world
        assert ranges[0][0] == 0 and sum(c for _, c in ranges) == len(costs)
        for (b0, c0), (b1, _) in zip(ranges[:-1], ranges[1:]):
            assert b0 + c0 == b1
        totals = [sum(costs[b:b + c]) for b, c in ranges]
        assert max(totals) <= sum(costs) / world + max(costs) + 1e-9


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
