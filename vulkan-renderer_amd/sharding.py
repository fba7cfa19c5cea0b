"""Host-side sharding of a bake across one process per GPU.

Every output texel of the Lambertian and GGX integrals (cube.cpp:95-133,240-313)
and every cell of the BRDF table (cube.cpp:346-388) is independent given the whole
input, so ranks take shares of the output and the only exchange step is one
all-gather of the finished shares per cube.

Two kinds of shares:
  * ranges of a linear order (BRDF-table cells, texels of the sampled GGX mode):
    shard_range / all_gather_ranges;
  * the library's shards of a Lambertian or GGX cube (ibl_*_shard_dev: groups of
    mirror-image texels, scattered over the cube): every rank packs its shard's texels
    (ibl_shard_pack_dev, or by index on the CPU), all_gather_packed moves them, and the
    packed shards are scattered into the cube (ibl_shard_unpack_dev / unpack_by_indices).
"""


def shard_range(n_units, rank, world_size):
    """Contiguous equal-size ranges (the last ones may be short or empty)."""
    per = (n_units + world_size - 1) // world_size
    begin = min(rank * per, n_units)
    return begin, min(per, n_units - begin)


def all_gather_ranges(dist, out_flat, n_units, rank, world_size, group=None, via_host=False):
    """All-gathers the per-rank ranges of `out_flat` ((n_units, C) tensor whose own
    range this rank has filled) so that every rank ends with the full tensor.
    One collective: ranges are padded to a common length.  via_host: see gather_cube."""
    import torch
    if world_size == 1:
        return out_flat
    per = (n_units + world_size - 1) // world_size
    begin, count = shard_range(n_units, rank, world_size)
    c = out_flat.shape[1]
    send = torch.zeros((per, c), dtype=out_flat.dtype, device=out_flat.device)
    if count:
        send[:count] = out_flat[begin:begin + count]
    recv = all_gather_packed(dist, send, world_size, group=group, via_host=via_host)
    out_flat.copy_(recv[:n_units])
    return out_flat


def all_gather_packed(dist, packed, world_size, group=None, via_host=False, always=False):
    """One all-gather of equal-size packed shards: (slots, C) per rank -> (world_size * slots, C),
    shard r at rows [r * slots, (r + 1) * slots).  via_host, always: see gather_cube."""
    import torch
    if world_size == 1 and not always:
        return packed
    send = packed.contiguous().cpu() if via_host else packed.contiguous()
    recv = torch.empty((world_size * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.to(packed.device) if via_host else recv


def gather_cube(dist, ibl, kind, in_w, cube, mip, mips, rank, world_size, group=None, via_host=False, always=False):
    """The exchange step of a sharded Lambertian / GGX cube on GPUs: pack this rank's shard of
    `cube`, all-gather, scatter every shard into `cube`.  via_host moves the packed shards
    through host tensors: for rehearsing several ranks on ONE GPU over gloo, whose staging of
    device tensors can hang when ranks share a device (the real path is RCCL on device
    tensors, one GPU per rank).  always: run the three steps for a single rank too (the way to
    put the RCCL collective itself under test on a one-GPU box)."""
    if world_size == 1 and not always:
        return cube
    packed = ibl.shard_pack(kind, in_w, cube, mip, mips, rank, world_size)
    every = all_gather_packed(dist, packed, world_size, group=group, via_host=via_host, always=always)
    return ibl.shard_unpack(kind, in_w, every, cube, mip, mips, world_size)


def pack_by_indices(cube_flat, indices):
    """CPU form of ibl_shard_pack_dev: cube_flat is (texels, C), indices the slots' texel
    numbers (0xffffffff = padding, packed as zeros)."""
    import torch
    idx = torch.as_tensor(indices.astype("int64"))
    valid = idx != 0xffffffff
    out = torch.zeros((idx.shape[0], cube_flat.shape[1]), dtype=cube_flat.dtype)
    out[valid] = cube_flat[idx[valid]]
    return out


def unpack_by_indices(packed_all, all_indices, cube_flat):
    """CPU form of ibl_shard_unpack_dev: all_indices is the concatenation of every shard's slots."""
    import torch
    idx = torch.as_tensor(all_indices.astype("int64"))
    valid = idx != 0xffffffff
    cube_flat[idx[valid]] = packed_all[valid]
    return cube_flat
