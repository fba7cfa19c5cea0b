// Lambertian irradiance through the cube's mirror symmetry (gfx950).
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
// mostly agrees on which, so usually 32 x 3 FMAs are issued, not 56 x 3.  The reference computes each of those 64 products with the
// same operands in the same order, so sharing them changes nothing it could see.
//
// Kernel plan
//   lambertian_oct_prep    one thread per input octet: canonical g and the 8
//                          members' radiance, channel-major (R[8], G[8], B[8]), packed
//                          as 7 float4 (112 B): a packed FMA then reads two neighbouring
//                          members of one channel, swapped by op_sel where the pattern
//                          flips x.
//   lambertian_oct_pairs   grid = (blocks of 256 output octets) x (input slices);
//                          a thread owns one output octet: 24 FP32 sums over one
//                          LDS tile of 64 input octets, then FP64 running sums.
//   lambertian_oct_reduce / lambertian_oct_final
//                          per output octet: slices summed in a fixed two-stage order in FP64,
//                          scaled, scattered to the 8 member texels that lie in the
//                          requested (face, y, x) range.
// A texel's value does not depend on the range or shard a call covers.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace ibl {

constexpr int OCT_TILE = 64;          // input octets per LDS tile (7 float4 each)
constexpr int OCT_THREADS = 256;
constexpr int OCT_REC = 7;            // float4 per input octet record

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

__global__ __launch_bounds__(256) void lambertian_oct_prep(const float4* __restrict__ env, uint32_t w, uint32_t h,
                                                            uint32_t n_oct, float4* __restrict__ rec) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_oct) return;
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

__device__ __forceinline__ void oct_decode(uint32_t g, uint32_t hw, uint32_t tw, uint32_t& axis, uint32_t& oj, uint32_t& oi) {
    const uint32_t t2 = tw * tw, tpr = hw / tw;
    const uint32_t tile = g / t2, within = g - tile * t2;
    const uint32_t ly = within / tw, lx = within - ly * tw;
    axis = tile / (tpr * tpr);
    const uint32_t trem = tile - axis * (tpr * tpr);
    const uint32_t ty = trem / tpr, tx = trem - ty * tpr;
    oj = ty * tw + ly;
    oi = tx * tw + lx;
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
__global__ __launch_bounds__(OCT_THREADS, OCT_MIN_WAVES) void lambertian_oct_pairs(
    const float4* __restrict__ rec, uint32_t n_oct_in, uint32_t slice_len,
    uint32_t out_size, uint32_t oct_begin, uint32_t oct_count, uint32_t out_begin, uint32_t out_count,
    double* __restrict__ partial) {
    __shared__ float4 s_rec[2][OCT_TILE * OCT_REC];

    const uint32_t tid = threadIdx.x;
    const uint32_t slice = blockIdx.y;
    const uint32_t in0 = slice * slice_len;
    const uint32_t in1 = min(n_oct_in, in0 + slice_len);
    const uint32_t n_tiles = (in1 - in0 + OCT_TILE - 1) / OCT_TILE;

    const uint32_t lo = blockIdx.x * OCT_THREADS + tid;              // local output octet
    const uint32_t hw = out_size / 2;
    const uint32_t oo = oct_begin + min(lo, oct_count - 1);
    uint32_t axis, oj, oi;
    oct_decode(oo, hw, oct_tile_edge(hw), axis, oj, oi);
    uint32_t mask = lo < oct_count ? octet_range_mask(axis, oj, oi, out_size, out_begin, out_count) : 0u;
    if (!__syncthreads_or(mask != 0u)) return;

    Texel c = octet_member(axis, oj, oi, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));   // cube.cpp:97

    // [channel][member pair]: the two members of a pair are neighbours in the record, so one packed
    // FMA serves both; a pattern that flips x pairs member so with so ^ 1, i.e. reads the record
    // pair with its halves swapped (an operand modifier of the packed FMA, not an instruction).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[3][4];
    double dacc[3][8];
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int p = 0; p < 4; p++) acc[k][p] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 8; s++) dacc[k][s] = 0.0;
    }
    const unsigned long long full = __ballot(true);     // the wave's live lanes

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
#pragma unroll 2
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
            // The other six patterns come in three pairs of opposite sign, (t, 7 - t): per lane
            // exactly one of a pair is positive (cube.cpp:115 drops the other).  When the whole
            // wave agrees on which, the other one's 24 FMAs are skipped; (-,-,-) never counts.
            {
                const float v = cc - s2;                                 // (-,+,+) = -a+b+c; (+,-,-) is -v
                const unsigned long long pos = __ballot(v > 0.0f);
                if (pos != 0ull) accumulate(std::integral_constant<int, 1>{}, sub_clamp01(cc, s2));
                if (pos != full) accumulate(std::integral_constant<int, 6>{}, sub_clamp01(s2, cc));
            }
            {
                const float v = s2 + cc;                                 // (+,-,+) = a-b+c; (-,+,-) is -v
                const unsigned long long pos = __ballot(v > 0.0f);
                if (pos != 0ull) accumulate(std::integral_constant<int, 2>{}, add_clamp01(s2, cc));
                if (pos != full) accumulate(std::integral_constant<int, 5>{}, neg_sub_clamp01(s2, cc));
            }
            {
                const float v = s1 - cc;                                 // (+,+,-) = a+b-c; (-,-,+) is -v
                const unsigned long long pos = __ballot(v > 0.0f);
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

    if (lo < oct_count) {
#pragma unroll
        for (int s = 0; s < 8; s++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                partial[((size_t)slice * 24 + (s * 3 + k)) * oct_count + lo] = dacc[k][s];
    }
}

// First stage of the slice sum: group g adds its run of consecutive slices, in order, into
// one more set of partial sums.  One thread per (output octet, member, group).
__global__ __launch_bounds__(256) void lambertian_oct_reduce(const double* __restrict__ partial, uint32_t n_slices,
                                                              uint32_t per_group, uint32_t oct_count,
                                                              double* __restrict__ reduced) {
    const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y, g = blockIdx.z;
    if (lo >= oct_count) return;
    const uint32_t s0 = g * per_group, s1 = min(n_slices, s0 + per_group);
    double v[3] = {0.0, 0.0, 0.0};
    for (uint32_t sl = s0; sl < s1; sl++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += partial[((size_t)sl * 24 + (s * 3 + k)) * oct_count + lo];
    }
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) reduced[((size_t)g * 24 + (s * 3 + k)) * oct_count + lo] = v[k];
}

// One thread per (output octet, member): coalesced over the octet index.
__global__ __launch_bounds__(256) void lambertian_oct_final(const double* __restrict__ partial, uint32_t n_slices,
                                                             uint32_t out_size, uint32_t oct_begin, uint32_t oct_count,
                                                             uint32_t out_begin, uint32_t out_count, double scale,
                                                             float4* __restrict__ out) {
    const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y;
    if (lo >= oct_count) return;
    const uint32_t hw = out_size / 2;
    const uint32_t oo = oct_begin + lo;
    uint32_t axis, oj, oi;
    oct_decode(oo, hw, oct_tile_edge(hw), axis, oj, oi);
    Texel t = octet_member(axis, oj, oi, s, out_size, out_size);
    const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
    if (o - out_begin >= out_count) return;
    double v[3] = {0.0, 0.0, 0.0};
    for (uint32_t sl = 0; sl < n_slices; sl++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += partial[((size_t)sl * 24 + (s * 3 + k)) * oct_count + lo];
    }
    out[o] = make_float4((float)(v[0] * scale), (float)(v[1] * scale), (float)(v[2] * scale), 0.0f);
}

// ---- launch plan -----------------------------------------------------------

LambertianOctPlan lambertian_oct_plan(uint32_t in_w, uint32_t out_size, uint32_t oct_count) {
    LambertianOctPlan pl;
    pl.n_oct_in = 3 * (in_w / 2) * (in_w / 2);
    // Slice length depends on the input size only (range/shard independence).
    uint32_t tiles = (pl.n_oct_in + OCT_TILE - 1) / OCT_TILE;
    // At most 512 slices: an eighth of the output blocks of the largest case (one of 8 GPUs)
    // then still fills a GPU three times over, and one GPU is 5 % faster than with 128.
    uint32_t max_slices = 512;
    if (const char* e = getenv("IBL_LAMB_OCT_SLICES")) max_slices = (uint32_t)atoi(e);   // tuning knob
    if (max_slices < 1) max_slices = 1;
    uint32_t tiles_per_slice = (tiles + max_slices - 1) / max_slices;
    if (tiles_per_slice < 2) tiles_per_slice = 2;
    pl.slice_len = tiles_per_slice * OCT_TILE;
    pl.n_slices = (pl.n_oct_in + pl.slice_len - 1) / pl.slice_len;
    pl.out_blocks = (oct_count + OCT_THREADS - 1) / OCT_THREADS;
    // many slices are summed in two stages (fixed grouping, so still one summation order)
    pl.reduce_groups = pl.n_slices > 32 ? (pl.n_slices + 31) / 32 : 1;
    pl.partial_bytes = (size_t)(pl.n_slices + (pl.reduce_groups > 1 ? pl.reduce_groups : 0)) * 24 * oct_count * sizeof(double);
    pl.rec_bytes = (size_t)pl.n_oct_in * OCT_REC * sizeof(float4);
    (void)out_size;
    return pl;
}

hipError_t lambertian_oct_launch(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 uint32_t oct_begin, uint32_t oct_count, uint32_t out_begin, uint32_t out_count,
                                 float* d_out, void* d_rec, double* d_partial, hipStream_t stream,
                                 hipEvent_t ev0, hipEvent_t ev1) {
    lambertian_oct_prep<<<(pl.n_oct_in + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, in_w, pl.n_oct_in,
                                                                       (float4*)d_rec);
    dim3 grid(pl.out_blocks, pl.n_slices);
    if (ev0) (void)hipEventRecord(ev0, stream);
    lambertian_oct_pairs<<<grid, OCT_THREADS, 0, stream>>>((const float4*)d_rec, pl.n_oct_in, pl.slice_len, out_size,
                                                           oct_begin, oct_count, out_begin, out_count, d_partial);
    if (ev1) (void)hipEventRecord(ev1, stream);
    const double scale = 4.0 * 0.31830988618379067154 / ((double)in_w * (double)in_w);
    const double* d_sums = d_partial;
    uint32_t n_sums = pl.n_slices;
    if (pl.reduce_groups > 1) {
        double* d_reduced = d_partial + (size_t)pl.n_slices * 24 * oct_count;
        const uint32_t per_group = (pl.n_slices + pl.reduce_groups - 1) / pl.reduce_groups;
        lambertian_oct_reduce<<<dim3((oct_count + 255) / 256, 8, pl.reduce_groups), 256, 0, stream>>>(
            d_partial, pl.n_slices, per_group, oct_count, d_reduced);
        d_sums = d_reduced;
        n_sums = pl.reduce_groups;
    }
    lambertian_oct_final<<<dim3((oct_count + 255) / 256, 8), 256, 0, stream>>>(d_sums, n_sums, out_size, oct_begin,
                                                                      oct_count, out_begin, out_count, scale,
                                                                      (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
