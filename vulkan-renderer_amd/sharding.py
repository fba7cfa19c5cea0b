"""Host-side sharding of a bake across one process per GPU.

Every output texel of the Lambertian and GGX integrals (cube.cpp:95-133,240-313)
and every cell of the BRDF table (cube.cpp:346-388) is independent given the whole
input, so ranks take contiguous ranges of the (face, y, x) output order and the
only exchange step is one all-gather of the finished ranges.
"""


def shard_range(n_units, rank, world_size):
    """Contiguous equal-size ranges (the last ones may be short or empty)."""
    per = (n_units + world_size - 1) // world_size
    begin = min(rank * per, n_units)
    return begin, min(per, n_units - begin)


def weighted_shard_ranges(costs, world_size):
    """Split units with per-unit costs into world_size contiguous ranges of roughly
    equal total cost.  Returns [(begin, count)] per rank."""
    total = float(sum(costs))
    ranges, begin, acc, k = [], 0, 0.0, 1
    for i, c in enumerate(costs):
        acc += c
        while k < world_size and acc >= total * k / world_size - 1e-12:
            ranges.append((begin, i + 1 - begin))
            begin = i + 1
            k += 1
    ranges.append((begin, len(costs) - begin))
    while len(ranges) < world_size:
        ranges.append((len(costs), 0))
    return ranges


def all_gather_ranges(dist, out_flat, n_units, rank, world_size, group=None):
    """All-gathers the per-rank ranges of `out_flat` ((n_units, C) tensor whose own
    range this rank has filled) so that every rank ends with the full tensor.
    One collective: ranges are padded to a common length."""
    import torch
    if world_size == 1:
        return out_flat
    per = (n_units + world_size - 1) // world_size
    begin, count = shard_range(n_units, rank, world_size)
    c = out_flat.shape[1]
    send = torch.zeros((per, c), dtype=out_flat.dtype, device=out_flat.device)
    if count:
        send[:count] = out_flat[begin:begin + count]
    recv = torch.empty((world_size * per, c), dtype=out_flat.dtype, device=out_flat.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    out_flat.copy_(recv[:n_units])
    return out_flat
