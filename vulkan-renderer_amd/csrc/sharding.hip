// The partition of a cube's output texels into shards, as ibl_lambertian_shard_dev and
// ibl_ggx_level_shard_dev use it, and the gather / scatter kernels that move a shard's texels
// between the full cube and a packed buffer (what an all-gather between GPUs carries).
//
// The partitions:
//   RANGE     contiguous (face, y, x) ranges                     (general kernels)
//   LAMB_OCT  contiguous ranges of the tile-major octet order, 8 texels per octet
//                                                                 (Lambertian octet kernel)
//   GGX_OCT   every n_shards-th unit of 64 octets of the tile-major order
//                                                                 (GGX octet kernel: cost per
//                                                                 unit varies over the cube)
//   GGX_HEX   every n_shards-th unit of 64 hexadecets (16 texels each; lanes and members that
//             do not exist are padding)                           (GGX 16-fold kernel)
//   LAMB_O48  every n_shards-th unit of 64 orbits of the cube's whole symmetry group (48 texels
//             each, 24 on the face diagonals)                     (Lambertian 48-fold kernel)
// A shard is a list of `per` slots (the same number for every shard, so that the packed
// buffers of all shards have one size); a slot names a texel or is padding.
#include "ibl_device.h"
#include "ibl_kernels.h"

namespace ibl {

uint32_t shard_slots(const ShardMap& m) {
    const uint32_t n_out = 6 * m.out_w * m.out_w;
    const uint32_t n_oct = 3 * (m.out_w / 2) * (m.out_w / 2);
    switch (m.partition) {
        case SHARD_LAMB_OCT: return (n_oct + m.n_shards - 1) / m.n_shards * 8;
        case SHARD_GGX_OCT: {
            const uint32_t n_units = (n_oct + GGX_OCT_UNIT - 1) / GGX_OCT_UNIT;
            return (n_units + m.n_shards - 1) / m.n_shards * GGX_OCT_UNIT * 8;
        }
        case SHARD_GGX_HEX:
            return (hex_units_total(m.out_w) + m.n_shards - 1) / m.n_shards * GH_UNIT * 16;
        case SHARD_LAMB_O48: {
            const uint32_t n_units = (o48_orbits_total(m.out_w) + GH_UNIT - 1) / GH_UNIT;
            return (n_units + m.n_shards - 1) / m.n_shards * GH_UNIT * 48;
        }
        default: return (n_out + m.n_shards - 1) / m.n_shards;
    }
}

// Texel index (face, y, x order) of slot `slot` of shard `shard`, or 0xffffffff for padding.
__host__ __device__ inline uint32_t shard_slot_texel(const ShardMap& m, uint32_t per, uint32_t shard, uint32_t slot) {
    const uint32_t w = m.out_w, hw = w / 2;
    const uint32_t n_out = 6 * w * w, n_oct = 3 * hw * hw;
    if (slot >= per) return 0xffffffffu;
    if (m.partition == SHARD_GGX_HEX) {
        const uint32_t unit = shard + (slot / (GH_UNIT * 16)) * m.n_shards;
        if (unit >= hex_units_total(w)) return 0xffffffffu;
        const HexLane h = hex_unit_lane(unit, (slot / 16) % GH_UNIT, hw);
        const uint32_t mem = slot % 16;
        if (!h.valid || (h.diagonal && mem >= 8u)) return 0xffffffffu;
        Texel t = hex_member(h.axis, h.j, h.i, mem, w);
        return (t.face * w + t.y) * w + t.x;
    }
    if (m.partition == SHARD_LAMB_O48) {
        const uint32_t unit = shard + (slot / (GH_UNIT * 48)) * m.n_shards;
        const O48Lane ol = o48_lane(unit * GH_UNIT + (slot / 48) % GH_UNIT, hw);
        const uint32_t mem = slot % 48;
        if (!ol.valid || (ol.diagonal && ((mem >> 3) & 1u))) return 0xffffffffu;
        Texel t = o48_member(ol.j, ol.i, mem, w);
        return (t.face * w + t.y) * w + t.x;
    }
    uint32_t oct;
    if (m.partition == SHARD_LAMB_OCT) {
        const uint32_t per_oct = per / 8;
        oct = shard * per_oct + slot / 8;
    } else if (m.partition == SHARD_GGX_OCT) {
        const uint32_t n_units = (n_oct + GGX_OCT_UNIT - 1) / GGX_OCT_UNIT;
        const uint32_t unit = shard + (slot / (GGX_OCT_UNIT * 8)) * m.n_shards;
        if (unit >= n_units) return 0xffffffffu;
        oct = unit * GGX_OCT_UNIT + (slot / 8) % GGX_OCT_UNIT;
    } else {
        const uint32_t o = shard * per + slot;
        return o < n_out ? o : 0xffffffffu;
    }
    if (oct >= n_oct) return 0xffffffffu;
    uint32_t axis, oj, oi;
    octet_decode(oct, hw, m.tw, axis, oj, oi);
    Texel t = octet_member(axis, oj, oi, slot % 8, w, w);
    return (t.face * w + t.y) * w + t.x;
}

void shard_indices(const ShardMap& m, uint32_t shard, uint32_t* indices) {
    const uint32_t per = shard_slots(m);
    for (uint32_t k = 0; k < per; k++) indices[k] = shard_slot_texel(m, per, shard, k);
}

template <typename T>
__global__ __launch_bounds__(256) void shard_pack(ShardMap m, uint32_t per, uint32_t shard, const T* __restrict__ full,
                                                   T* __restrict__ packed) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= per) return;
    const uint32_t o = shard_slot_texel(m, per, shard, k);
    T zero = {};
    packed[k] = o != 0xffffffffu ? full[o] : zero;
}

template <typename T>
__global__ __launch_bounds__(256) void shard_unpack(ShardMap m, uint32_t per, const T* __restrict__ packed_all,
                                                     T* __restrict__ full) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t shard = blockIdx.y;
    if (k >= per) return;
    const uint32_t o = shard_slot_texel(m, per, shard, k);
    if (o != 0xffffffffu) full[o] = packed_all[(size_t)shard * per + k];
}

hipError_t shard_pack_launch(const ShardMap& m, uint32_t shard, uint32_t elem_bytes, const void* d_full, void* d_packed,
                             hipStream_t stream) {
    const uint32_t per = shard_slots(m);
    if (per == 0) return hipSuccess;
    const dim3 grid((per + 255) / 256);
    if (elem_bytes == 16) shard_pack<float4><<<grid, 256, 0, stream>>>(m, per, shard, (const float4*)d_full, (float4*)d_packed);
    else shard_pack<uint32_t><<<grid, 256, 0, stream>>>(m, per, shard, (const uint32_t*)d_full, (uint32_t*)d_packed);
    return hipGetLastError();
}

hipError_t shard_unpack_launch(const ShardMap& m, uint32_t elem_bytes, const void* d_packed_all, void* d_full,
                               hipStream_t stream) {
    const uint32_t per = shard_slots(m);
    if (per == 0) return hipSuccess;
    const dim3 grid((per + 255) / 256, m.n_shards);
    if (elem_bytes == 16) shard_unpack<float4><<<grid, 256, 0, stream>>>(m, per, (const float4*)d_packed_all, (float4*)d_full);
    else shard_unpack<uint32_t><<<grid, 256, 0, stream>>>(m, per, (const uint32_t*)d_packed_all, (uint32_t*)d_full);
    return hipGetLastError();
}

}  // namespace ibl
