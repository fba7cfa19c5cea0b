// Lambertian irradiance through the cube's mirror symmetry, with the reference's own per-pair
// weights (gfx950).  lambertian_split.hip evaluates the same sum with a quarter of the
// arithmetic and is what runs first; the kernels of this file then redo the output octets whose
// texels that form cannot hold to ~1e-6 (or everything, with the knob "lamb_split" at 0).
//
// Same sum as lambertian.hip (integrateLambertian, src/utilities/cube.cpp:95-129),
// evaluated octet against octet.  With canonical (all-positive) components of an
// output direction n and of a folded input direction g = w/|w|^4, the dot product
// of output member so with input member si is  tx*a + ty*b + tz*c  with
// a = nx*gx, b = ny*gy, c = nz*gz and t = so (x) si (component-wise sign product).
// So 3 multiplies, 2 adds and 6 clamped adds give all 64 dot products of an
// (output octet, input octet) pair (the pattern t = (-,-,-) is never positive),
// and what remains is the 56 x 3 colour FMAs: 180 VALU instructions per 64 texel
// pairs instead of 384.  Of the six remaining non-trivial patterns, opposite ones
// cancel: per lane exactly three are positive, and a wave of neighbouring octets
// mostly agrees on which, so usually 32 x 3 FMAs are issued, not 56 x 3.  The reference
// computes each of those 64 products with the same operands in the same order, so sharing
// them changes nothing it could see.
//
// Kernel plan
//   lambertian_oct_prep    one thread per input octet: canonical g and the 8
//                          members' radiance, channel-major (R[8], G[8], B[8]), packed
//                          as 7 float4 (112 B): a packed FMA then reads two neighbouring
//                          members of one channel, swapped by op_sel where the pattern
//                          flips x.
//   lambertian_oct_pairs   grid = (blocks of 256 output octets) x (input slices, dealt to the
//                          XCDs); a thread owns one output octet: 24 FP32 sums over one
//                          LDS tile of 64 input octets, FP64 sums over the slice, stored as
//                          FP32 slice sums (or, in the run form, FP64 sums of 8 slices).
//   lambertian_oct_reduce / lambertian_oct_final
//                          per output octet: slices summed in a fixed two-stage order in FP64,
//                          scaled, scattered to the 8 member texels that lie in the
//                          requested (face, y, x) range.
// A texel's value does not depend on the range or shard a call covers.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace ibl {

constexpr int OCT_TILE = 64;          // input octets per LDS tile (7 float4 each)
constexpr int OCT_THREADS = 256;
constexpr int OCT_REC = 7;            // float4 per input octet record
constexpr int OCT_RUN = 8;            // slices per run: the first stage of the fixed summation order

// s(i) = -s(w-1-i) exactly for every column: true for every power of two, checked
// for the rest (getNormalFrom evaluates s in double and rounds once, cube.cpp:32).
bool cube_is_mirror_exact(uint32_t w) {
    if (w < 2 || (w & 1u)) return false;
    for (uint32_t i = 0; i < w / 2; i++) {
        volatile float a = (float)((((double)(float)i + 0.5) / (double)(float)w) * 2.0 - 1.0);
        volatile float b = (float)((((double)(float)(w - 1 - i) + 0.5) / (double)(float)w) * 2.0 - 1.0);
        if (!(a == -b) || !(a < 0.0f)) return false;
    }
    return true;
}

// any_flag (here and below: flags, one per output octet of the call, then one per 256-octet
// block): when given, only the blocks that hold a marked octet are worked on, and only the marked
// octets are stored -- the second pass after lambertian_split.hip; nothing at all happens when no
// octet is marked.
__global__ __launch_bounds__(256) void lambertian_oct_prep(const float4* __restrict__ env, uint32_t w, uint32_t h,
                                                            uint32_t n_oct, const uint32_t* __restrict__ any_flag,
                                                            float4* __restrict__ rec) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_oct) return;
    if (any_flag && *any_flag == 0u) return;
    const uint32_t hw = w / 2, hh = h / 2;
    uint32_t axis = k / (hw * hh);
    uint32_t rem = k - axis * (hw * hh);
    uint32_t j = rem / hw, i = rem - j * hw;
    Texel c = octet_member(axis, j, i, 0, w, h);
    F3 d = ref_normal_from(c.face, c.x, c.y, w, h);       // all components > 0
    float l2 = ref_dot(d, d);
    float inv = 1.0f / (l2 * l2);
    float v[28];
    v[0] = d.x * inv; v[1] = d.y * inv; v[2] = d.z * inv;
#pragma unroll
    for (uint32_t s = 0; s < 8; s++) {
        Texel t = octet_member(axis, j, i, s, w, h);
        float4 L = env[(t.face * h + t.y) * h + t.x];     // cube.cpp:104 row stride
        v[4 + s] = L.x; v[12 + s] = L.y; v[20 + s] = L.z;      // channel-major: R[8], G[8], B[8]
    }
    v[3] = 0.0f;
#pragma unroll
    for (int q = 0; q < OCT_REC; q++)
        rec[(size_t)k * OCT_REC + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// Output octets are numbered tile-major (tiles of tw x tw canonical positions, tw <= 8), so
// that the 64 octets of a wave are neighbours on the cube: what a wave can skip below
// depends on how far apart its directions are.
__host__ __device__ inline uint32_t oct_tile_edge(uint32_t hw) {
    uint32_t tw = hw < 8 ? hw : 8;
    while (hw % tw) tw--;
    return tw;
}

__host__ __device__ inline void oct_decode(uint32_t g, uint32_t hw, uint32_t tw, uint32_t& axis, uint32_t& oj, uint32_t& oi) {
    octet_decode(g, hw, tw, axis, oj, oi);
}

// Bit mask of the members of output octet (axis, j, i) whose (face, y, x) index
// lies in [out_begin, out_begin + out_count).
__device__ __forceinline__ uint32_t octet_range_mask(uint32_t axis, uint32_t j, uint32_t i, uint32_t w,
                                                      uint32_t out_begin, uint32_t out_count) {
    uint32_t m = 0;
#pragma unroll
    for (uint32_t s = 0; s < 8; s++) {
        Texel t = octet_member(axis, j, i, s, w, w);
        uint32_t o = (t.face * w + t.y) * w + t.x;
        if (o - out_begin < out_count) m |= 1u << s;      // unsigned wrap handles o < out_begin
    }
    return m;
}

#ifndef OCT_MIN_WAVES
#define OCT_MIN_WAVES 4   // measured: 4 is 3-4 % faster than 2-3, 5-6 are 10 % slower
#endif
#ifndef OCT_RUN_UNROLL
#define OCT_RUN_UNROLL 1
#endif
#ifndef OCT_MIN_WAVES_RUN
#define OCT_MIN_WAVES_RUN 3   // the run form carries 24 more FP64 sums per thread
#endif

// Summation order of one output sum (fixed by the input size alone, so that a texel's bits
// do not depend on how a bake is split into calls):
//   slice sum  S_s  = FP64 running sum, from zero, of the FP32 sums of the slice's tiles in order,
//                     rounded to FP32 once (3e-8 of one of several hundred slices: ~1e-9 of the total);
//   run sum    B_b  = ((S_8b + S_8b+1) + ...) in FP64 over the run's OCT_RUN consecutive slices in order;
//   total           = ((B_0 + B_1) + ...) in FP64.
// RUN = false: a workgroup covers one slice and stores S_s (4 bytes a sum); lambertian_oct_reduce
//              forms the B_b.  The default: 4 waves per SIMD.
// RUN = true : a workgroup walks one whole run and stores B_b: a sixteenth of the bytes leave the
//              chip, but the 24 extra FP64 sums cost a wave per SIMD and 12 % of the speed (measured
//              on 1024 -> 128: 22.1 against 19.6 ms).  Used where the workspace must be small.
// Both give the same bits.
// COUNT: the wavefronts also count the records they walk and the sign patterns they evaluate
// (what the arithmetic actually executed is priced by).
template <bool RUN, bool COUNT>
__global__ __launch_bounds__(OCT_THREADS, RUN ? OCT_MIN_WAVES_RUN : OCT_MIN_WAVES) void lambertian_oct_pairs(
    const float4* __restrict__ rec, uint32_t n_oct_in, uint32_t slice_len, uint32_t n_slices,
    uint32_t out_size, const uint32_t* __restrict__ block_list, uint32_t oct_begin, uint32_t oct_end,
    uint32_t out_blocks, uint32_t xcd_affine, uint32_t out_begin, uint32_t out_count,
    const uint32_t* __restrict__ flags, void* __restrict__ partial_out, unsigned long long* __restrict__ counts) {
    __shared__ float4 s_rec[2][OCT_TILE * OCT_REC];

    const uint32_t tid = threadIdx.x;
    // Workgroups go to the 8 XCDs round-robin by their linear index and every XCD has its own
    // L2.  With xcd_affine XCD k only ever works on slots k, k + 8, ...: every input record is
    // then fetched into one L2 instead of all eight (the grid is padded to a multiple of 8 slots;
    // workgroups of the padding leave at once).
    uint32_t slot, ob;
    if (xcd_affine) {
        const uint32_t seq = blockIdx.x >> 3;
        slot = (blockIdx.x & 7u) + 8u * (seq / out_blocks);
        ob = seq % out_blocks;
        if (slot >= (RUN ? (n_slices + OCT_RUN - 1) / OCT_RUN : n_slices)) return;
    } else {
        slot = blockIdx.x / out_blocks;
        ob = blockIdx.x % out_blocks;
    }
    if (flags && flags[out_blocks * OCT_THREADS + ob] == 0u) return;
    const uint32_t sl0 = RUN ? slot * OCT_RUN : slot;
    const uint32_t sl1 = RUN ? min(n_slices, sl0 + OCT_RUN) : sl0 + 1;

    const uint32_t lo = ob * OCT_THREADS + tid;                      // local output octet
    const uint32_t oct_local = out_blocks * OCT_THREADS;
    const uint32_t hw = out_size / 2;
    const uint32_t oo_raw = (block_list ? block_list[ob] * OCT_THREADS : oct_begin + ob * OCT_THREADS) + tid;
    const bool valid = oo_raw < oct_end;
    const uint32_t oo = valid ? oo_raw : oct_end - 1;
    uint32_t axis, oj, oi;
    oct_decode(oo, hw, oct_tile_edge(hw), axis, oj, oi);
    uint32_t mask = valid ? octet_range_mask(axis, oj, oi, out_size, out_begin, out_count) : 0u;
    if (!__syncthreads_or(mask != 0u)) return;

    Texel c = octet_member(axis, oj, oi, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));   // cube.cpp:97

    // [channel][member pair]: the two members of a pair are neighbours in the record, so one packed
    // FMA serves both; a pattern that flips x pairs member so with so ^ 1, i.e. reads the record
    // pair with its halves swapped (an operand modifier of the packed FMA, not an instruction).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[3][4];
    double dacc[3][8];
    double gacc[RUN ? 3 : 1][RUN ? 8 : 1];
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int p = 0; p < 4; p++) acc[k][p] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 8; s++) {
            dacc[k][s] = 0.0;
            if (RUN) gacc[k][s] = 0.0;
        }
    }
    const unsigned long long full = __ballot(true);     // the wave's live lanes
    unsigned long long n_records = 0, n_patterns = 0;    // wave-uniform

    for (uint32_t slice = sl0; slice < sl1; slice++) {
        const uint32_t in0 = slice * slice_len;
        const uint32_t in1 = min(n_oct_in, in0 + slice_len);
        const uint32_t n_tiles = (in1 - in0 + OCT_TILE - 1) / OCT_TILE;

        // Tile staging: OCT_TILE*OCT_REC = 448 float4 per tile, two per thread.
        auto fetch = [&](uint32_t tile, float4& v0, float4& v1) {
            const size_t base = ((size_t)in0 + (size_t)tile * OCT_TILE) * OCT_REC;
            const uint32_t lim = (min(in1 - (in0 + tile * OCT_TILE), (uint32_t)OCT_TILE)) * OCT_REC;
            v0 = tid < lim ? rec[base + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
            v1 = (tid + OCT_THREADS < lim) ? rec[base + tid + OCT_THREADS] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        float4 v0, v1;
        fetch(0, v0, v1);
        s_rec[0][tid] = v0;
        if (tid + OCT_THREADS < OCT_TILE * OCT_REC) s_rec[0][tid + OCT_THREADS] = v1;
        __syncthreads();

        for (uint32_t tile = 0; tile < n_tiles; tile++) {
            const int cur = tile & 1;
            if (tile + 1 < n_tiles) fetch(tile + 1, v0, v1);
#pragma unroll RUN ? OCT_RUN_UNROLL : 2     // the run form has no registers to spare for a second record in flight
            for (int e = 0; e < OCT_TILE; e++) {
                const float4 g4 = s_rec[cur][e * OCT_REC];
                f32x2 rp[3][4];          // radiance: [channel][member pair]
#pragma unroll
                for (int q = 1; q < OCT_REC; q++) {
                    const float4 t4 = s_rec[cur][e * OCT_REC + q];
                    rp[(q - 1) / 2][2 * ((q - 1) % 2)] = f32x2{t4.x, t4.y};
                    rp[(q - 1) / 2][2 * ((q - 1) % 2) + 1] = f32x2{t4.z, t4.w};
                }
                const float a = n.x * g4.x, b = n.y * g4.y, cc = n.z * g4.z;
                const float s1 = a + b, s2 = a - b;
                // one sign pattern: output member so pairs with input member so ^ TAU
                auto accumulate = [&](auto tau_c, float t) {
                    constexpr int TAU = decltype(tau_c)::value;
                    const f32x2 t2 = f32x2{t, t};
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int p = 0; p < 4; p++) {
                            const f32x2 src = rp[k][p ^ (TAU >> 1)];
                            acc[k][p] = __builtin_elementwise_fma(t2, (TAU & 1) ? src.yx : src, acc[k][p]);
                        }
                };
                accumulate(std::integral_constant<int, 0>{}, s1 + cc);      // (+,+,+): always positive
                if (COUNT) { n_records += 1; n_patterns += 1; }
                // The other six patterns come in three pairs of opposite sign, (t, 7 - t): per lane
                // exactly one of a pair is positive (cube.cpp:115 drops the other).  When the whole
                // wave agrees on which, the other one's 24 FMAs are skipped; (-,-,-) never counts.
                {
                    const float v = cc - s2;                                 // (-,+,+) = -a+b+c; (+,-,-) is -v
                    const unsigned long long pos = __ballot(v > 0.0f);
                    if (COUNT) n_patterns += (pos != 0ull ? 1u : 0u) + (pos != full ? 1u : 0u);
                    if (pos != 0ull) accumulate(std::integral_constant<int, 1>{}, sub_clamp01(cc, s2));
                    if (pos != full) accumulate(std::integral_constant<int, 6>{}, sub_clamp01(s2, cc));
                }
                {
                    const float v = s2 + cc;                                 // (+,-,+) = a-b+c; (-,+,-) is -v
                    const unsigned long long pos = __ballot(v > 0.0f);
                    if (COUNT) n_patterns += (pos != 0ull ? 1u : 0u) + (pos != full ? 1u : 0u);
                    if (pos != 0ull) accumulate(std::integral_constant<int, 2>{}, add_clamp01(s2, cc));
                    if (pos != full) accumulate(std::integral_constant<int, 5>{}, neg_sub_clamp01(s2, cc));
                }
                {
                    const float v = s1 - cc;                                 // (+,+,-) = a+b-c; (-,-,+) is -v
                    const unsigned long long pos = __ballot(v > 0.0f);
                    if (COUNT) n_patterns += (pos != 0ull ? 1u : 0u) + (pos != full ? 1u : 0u);
                    if (pos != 0ull) accumulate(std::integral_constant<int, 4>{}, sub_clamp01(s1, cc));
                    if (pos != full) accumulate(std::integral_constant<int, 3>{}, sub_clamp01(cc, s1));
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    dacc[k][2 * p] += (double)acc[k][p].x;
                    dacc[k][2 * p + 1] += (double)acc[k][p].y;
                    acc[k][p] = f32x2{0.0f, 0.0f};
                }
            if (tile + 1 < n_tiles) {
                s_rec[cur ^ 1][tid] = v0;
                if (tid + OCT_THREADS < OCT_TILE * OCT_REC) s_rec[cur ^ 1][tid + OCT_THREADS] = v1;
            }
            __syncthreads();
        }
        if (RUN) {
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int s = 0; s < 8; s++) { gacc[k][s] += (double)(float)dacc[k][s]; dacc[k][s] = 0.0; }
        }
    }

    if (valid) {
#pragma unroll
        for (int s = 0; s < 8; s++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const size_t at = ((size_t)slot * 24 + (s * 3 + k)) * oct_local + lo;
                if (RUN) ((double*)partial_out)[at] = gacc[k][s];
                else ((float*)partial_out)[at] = (float)dacc[k][s];
            }
    }
    if (COUNT && (tid & 63u) == 0u) {
        atomicAdd(&counts[3], n_records);
        atomicAdd(&counts[4], n_patterns);
    }
}

// The run sums B_b from the slice sums S_s (RUN = false calls only): one thread per
// (output octet, member, run).
__global__ __launch_bounds__(256) void lambertian_oct_reduce(const float* __restrict__ partial, uint32_t n_slices,
                                                              uint32_t oct_local, const uint32_t* __restrict__ flags,
                                                              double* __restrict__ reduced) {
    const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y, g = blockIdx.z;
    if (lo >= oct_local) return;
    if (flags && flags[lo] == 0u) return;
    const uint32_t s0 = g * OCT_RUN, s1 = min(n_slices, s0 + OCT_RUN);
    double v[3] = {0.0, 0.0, 0.0};
    for (uint32_t sl = s0; sl < s1; sl++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += (double)partial[((size_t)sl * 24 + (s * 3 + k)) * oct_local + lo];
    }
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) reduced[((size_t)g * 24 + (s * 3 + k)) * oct_local + lo] = v[k];
}

// One thread per (output octet, member): coalesced over the octet index.
__global__ __launch_bounds__(256) void lambertian_oct_final(const double* __restrict__ runs, uint32_t n_runs,
                                                             uint32_t out_size, const uint32_t* __restrict__ block_list,
                                                             uint32_t oct_begin, uint32_t oct_end, uint32_t oct_local,
                                                             uint32_t out_begin, uint32_t out_count, double scale,
                                                             const uint32_t* __restrict__ flags,
                                                             float4* __restrict__ out) {
    const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y;
    if (lo >= oct_local) return;
    const uint32_t ob = lo / OCT_THREADS;
    if (flags && flags[lo] == 0u) return;
    const uint32_t oo = (block_list ? block_list[ob] * OCT_THREADS : oct_begin + ob * OCT_THREADS) + lo % OCT_THREADS;
    if (oo >= oct_end) return;
    const uint32_t hw = out_size / 2;
    uint32_t axis, oj, oi;
    oct_decode(oo, hw, oct_tile_edge(hw), axis, oj, oi);
    Texel t = octet_member(axis, oj, oi, s, out_size, out_size);
    const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
    if (o - out_begin >= out_count) return;
    double v[3] = {0.0, 0.0, 0.0};
    for (uint32_t b = 0; b < n_runs; b++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += runs[((size_t)b * 24 + (s * 3 + k)) * oct_local + lo];
    }
    out[o] = make_float4((float)(v[0] * scale), (float)(v[1] * scale), (float)(v[2] * scale), 0.0f);
}

// ---- launch plan -----------------------------------------------------------

uint32_t lambertian_oct_blocks_total(uint32_t out_size) {
    const uint32_t n_oct = 3 * (out_size / 2) * (out_size / 2);
    return (n_oct + OCT_THREADS - 1) / OCT_THREADS;
}

uint32_t lambertian_oct_tile_edge(uint32_t out_size) { return oct_tile_edge(out_size / 2); }

// ---- unit lists -------------------------------------------------------------
// A call that covers only part of a cube works on a list of units (blocks of `unit`
// consecutive octets).  The list is made on the device, in ascending order; the host counts
// with the same predicate, so it knows the length without reading anything back.

uint32_t octet_units_in_range_count(uint32_t unit, uint32_t out_w, uint32_t tw, uint32_t begin, uint32_t count) {
    const uint32_t hw = out_w / 2, n_oct = 3 * hw * hw, n_units = (n_oct + unit - 1) / unit;
    uint32_t n = 0;
    for (uint32_t u = 0; u < n_units; u++) n += octet_unit_in_range(u, unit, hw, tw, begin, count) ? 1u : 0u;
    return n;
}

__global__ __launch_bounds__(1024) void octet_units_in_range(uint32_t unit, uint32_t hw, uint32_t tw, uint32_t n_units,
                                                              uint32_t begin, uint32_t count, uint32_t* __restrict__ list) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t u0 = 0; u0 < n_units; u0 += 1024) {
        const uint32_t u = u0 + tid;
        const bool hit = u < n_units && octet_unit_in_range(u, unit, hw, tw, begin, count);
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_base;
        for (uint32_t k = 0; k < wave; k++) before += s_wave[k];
        if (hit) list[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = u;
        __syncthreads();
        if (tid == 0) {
            uint32_t total = 0;
            for (uint32_t k = 0; k < 16; k++) total += s_wave[k];
            s_base += total;
        }
        __syncthreads();
    }
}

hipError_t octet_units_in_range_launch(uint32_t unit, uint32_t out_w, uint32_t tw, uint32_t begin, uint32_t count,
                                       uint32_t* d_list, hipStream_t stream) {
    const uint32_t hw = out_w / 2, n_oct = 3 * hw * hw, n_units = (n_oct + unit - 1) / unit;
    octet_units_in_range<<<1, 1024, 0, stream>>>(unit, hw, tw, n_units, begin, count, d_list);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void octet_units_strided(uint32_t first, uint32_t stride, uint32_t n, uint32_t* __restrict__ list) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) list[k] = first + k * stride;
}

hipError_t octet_units_strided_launch(uint32_t first, uint32_t stride, uint32_t n, uint32_t* d_list, hipStream_t stream) {
    if (n) octet_units_strided<<<(n + 255) / 256, 256, 0, stream>>>(first, stride, n, d_list);
    return hipGetLastError();
}

// [run][24][octets] double, followed in the slice form by [slice][24][octets] float.  With the
// split form in front (lambertian_split.hip) the workspace starts with its header and holds
// whichever of the two passes' sums is larger.
size_t lambertian_oct_partial_bytes(const LambertianOctPlan& pl) {
    const size_t sums = (size_t)24 * pl.out_blocks * OCT_THREADS;
    const size_t exact = sums * sizeof(double) * pl.n_runs + (pl.run ? 0 : sums * sizeof(float) * pl.n_slices);
    if (!pl.split) return exact;
    const size_t fast = lambertian_split_sums_bytes(pl);
    return lambertian_split_header_bytes(pl.n_oct_in, pl.out_blocks) + (fast > exact ? fast : exact);
}

LambertianOctPlan lambertian_oct_plan(uint32_t in_w, uint32_t out_blocks) {
    LambertianOctPlan pl;
    pl.n_oct_in = 3 * (in_w / 2) * (in_w / 2);
    // Slice length depends on the input size only (range/shard independence).
    uint32_t tiles = (pl.n_oct_in + OCT_TILE - 1) / OCT_TILE;
    // At most 341 slices, 430 from 1024 x 1024 faces on (424 for a 1024 x 1024 face).  The count fixes the
    // order of the sums, so it may depend on the input size only; it is chosen for the 1024 -> 128 cube
    // on 1, 2, 4 and 8 GPUs at once.  An eighth of that cube is 6 output blocks: with 333 slices its
    // 1998 workgroups fill the 2048 a GPU holds (8 per CU) just once, and that single round takes
    // 0.65 ms where an eighth of the whole cube's time is 0.48; with 424 slices (2544 workgroups) it
    // takes 0.55 ms, with 512 or 683 no less, while the whole cube on one GPU stays at 3.8 ms
    // (tools/lamb_slices_sweep.py; starting the waves of a SIMD a hundred cycles apart changes nothing).
    // The slice sums grow with the count: 250 MB written for 153 MB the kernel has to move anyway.
    // Smaller inputs keep 341, and with it the sizes the static entry point's 64 MiB array holds.
    uint32_t max_slices = tuning().lamb_oct_slices ? tuning().lamb_oct_slices : (tiles >= 12288u ? 430u : 341u);
    uint32_t tiles_per_slice = (tiles + max_slices - 1) / max_slices;
    if (tiles_per_slice < 2) tiles_per_slice = 2;
    pl.slice_len = tiles_per_slice * OCT_TILE;
    pl.n_slices = (pl.n_oct_in + pl.slice_len - 1) / pl.slice_len;
    pl.n_runs = (pl.n_slices + OCT_RUN - 1) / OCT_RUN;
    pl.out_blocks = out_blocks;
    pl.run = tuning().lamb_oct_run > 0 && pl.n_slices > 1;
    pl.split = tuning().lamb_split != 0;
    pl.partial_bytes = lambertian_oct_partial_bytes(pl);
    pl.rec_bytes = (size_t)pl.n_oct_in * OCT_REC * sizeof(float4);
    return pl;
}

// The kernels of this file over the octets of a call; d_flags (or null: every octet) as described at
// lambertian_oct_prep, d_partial the sums workspace behind whatever header the pass in front keeps.
hipError_t lambertian_oct_exact_pass(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                     const uint32_t* d_block_list, uint32_t oct_begin, uint32_t oct_end,
                                     uint32_t out_begin, uint32_t out_count, float* d_out, void* d_rec, double* d_partial,
                                     const uint32_t* d_flags, unsigned long long* counts,
                                     hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t oct_local_all = pl.out_blocks * OCT_THREADS;
    lambertian_oct_prep<<<(pl.n_oct_in + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, in_w, pl.n_oct_in,
                                                                       d_flags ? d_flags + oct_local_all + pl.out_blocks : nullptr,
                                                                       (float4*)d_rec);
    const uint32_t oct_local = pl.out_blocks * OCT_THREADS;
    const uint32_t n_slots = pl.run ? pl.n_runs : pl.n_slices;
    const uint32_t affine = (n_slots >= 8 && tuning().lamb_oct_affine != 0) ? 1u : 0u;
    const dim3 grid(pl.out_blocks * (affine ? (n_slots + 7) / 8 * 8 : n_slots));
    if (ev0) (void)hipEventRecord(ev0, stream);
    // workspace: the run sums first, then (slice form) the slice sums
    double* d_runs = d_partial;
    float* d_slices = (float*)(d_partial + (size_t)pl.n_runs * 24 * oct_local);
    auto go = [&](auto kernel, void* d_store) {
        kernel<<<grid, OCT_THREADS, 0, stream>>>(
            (const float4*)d_rec, pl.n_oct_in, pl.slice_len, pl.n_slices, out_size, d_block_list, oct_begin, oct_end,
            pl.out_blocks, affine, out_begin, out_count, d_flags, d_store, counts);
    };
    if (pl.run) { if (counts) go(lambertian_oct_pairs<true, true>, d_runs); else go(lambertian_oct_pairs<true, false>, d_runs); }
    else        { if (counts) go(lambertian_oct_pairs<false, true>, d_slices); else go(lambertian_oct_pairs<false, false>, d_slices); }
    if (ev1) (void)hipEventRecord(ev1, stream);
    const double scale = 4.0 * 0.31830988618379067154 / ((double)in_w * (double)in_w);
    if (!pl.run)
        lambertian_oct_reduce<<<dim3((oct_local + 255) / 256, 8, pl.n_runs), 256, 0, stream>>>(
            (const float*)d_slices, pl.n_slices, oct_local, d_flags, d_runs);
    lambertian_oct_final<<<dim3((oct_local + 255) / 256, 8), 256, 0, stream>>>(
        d_runs, pl.n_runs, out_size, d_block_list, oct_begin, oct_end, oct_local, out_begin, out_count, scale, d_flags,
        (float4*)d_out);
    return hipGetLastError();
}

hipError_t lambertian_oct_launch(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 const uint32_t* d_block_list, uint32_t oct_begin, uint32_t oct_end,
                                 uint32_t out_begin, uint32_t out_count,
                                 float* d_out, void* d_rec, double* d_partial, unsigned long long* d_counts,
                                 hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3) {
    // Split form first: every block at a quarter of the arithmetic, then this file's kernels only
    // for the blocks it marked (none for an environment of ordinary contrast: the launches below
    // then return at once).
    const uint32_t* d_flags = nullptr;
    if (pl.split) {
        if (hipError_t e = lambertian_split_launch(pl, d_env, in_w, out_size, d_block_list, oct_begin, oct_end, out_begin,
                                                   out_count, d_out, d_rec, d_partial, d_counts, stream, ev0, ev1);
            e != hipSuccess) return e;
        d_flags = (const uint32_t*)d_partial;
        d_partial = (double*)((char*)d_partial + lambertian_split_header_bytes(pl.n_oct_in, pl.out_blocks));
        ev0 = ev2;      // this pass's pair kernel is timed by the call's second pair of events
        ev1 = ev3;
    }
    // (the second pass does not count: the counters price the kernel that does the work)
    return lambertian_oct_exact_pass(pl, d_env, in_w, out_size, d_block_list, oct_begin, oct_end, out_begin, out_count, d_out,
                                     d_rec, d_partial, d_flags, pl.split ? nullptr : d_counts, stream, ev0, ev1);
}

}  // namespace ibl
