// Lambertian irradiance convolution for gfx950: the all-pairs sum of
// integrateLambertian, src/utilities/cube.cpp:95-129.
//
//   E(n) = 4/(pi W H) * sum_j L_j * max(0, n.w_j) / |w_j|^4
//
// Kernel plan
//   lambertian_prep   one thread per input texel: folds the per-texel factor into
//                     the direction, g_j = w_j/|w_j|^4, and packs (g_j, L_j) as
//                     24 bytes (float4 + float2) so the pair loop is dot + 3 FMAs.
//   lambertian_pairs  grid = (blocks of 256*P output texels) x (input slices).
//                     Each thread keeps P output directions and 3P FP32 partial
//                     sums in registers; input texels stream HBM -> LDS in tiles
//                     of TILE texels and are read back as wave-uniform broadcasts.
//                     The FP32 sums cover one tile only (fixed order), then are
//                     added to FP64 running sums, which keeps the total within
//                     ~1e-7 relative of the reference's all-FP64 accumulation.
//   lambertian_final  sums the slices of every output texel in slice order in
//                     FP64, applies 4/(pi W H) and stores float RGBA (alpha 0).
// The result for an output texel does not depend on how many texels the call
// covers (the slice length is a function of the input size only), so a sharded
// bake is bit-identical to an unsharded one.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include <stdlib.h>

namespace ibl {

constexpr int LAMB_TILE = 256;      // input texels per LDS tile == FP32 chunk length
constexpr int LAMB_THREADS = 256;

__global__ __launch_bounds__(256) void lambertian_prep(const float4* __restrict__ env, uint32_t w, uint32_t h,
                                                        uint32_t n_in, float4* __restrict__ a, float2* __restrict__ b) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_in) return;
    uint32_t f = j / (w * h);
    uint32_t rem = j - f * (w * h);
    uint32_t y = rem / w;
    uint32_t x = rem - y * w;
    F3 d = ref_normal_from(f, x, y, w, h);
    float l2 = ref_dot(d, d);
    float inv = 1.0f / (l2 * l2);
    // cube.cpp:104 strides input rows by the height; faces are square.
    float4 L = env[(f * h + y) * h + x];
    a[j] = make_float4(d.x * inv, d.y * inv, d.z * inv, L.x);
    b[j] = make_float2(L.y, L.z);
}

#ifndef IBL_LAMB_MIN_WAVES
#define IBL_LAMB_MIN_WAVES 2
#endif

template <int P>
__global__ __launch_bounds__(LAMB_THREADS, IBL_LAMB_MIN_WAVES) void lambertian_pairs(
    const float4* __restrict__ a, const float2* __restrict__ b, uint32_t n_in, uint32_t slice_len,
    uint32_t out_size, uint32_t out_begin, uint32_t out_count, double* __restrict__ partial) {
    __shared__ float4 s_a[2][LAMB_TILE];
    __shared__ float2 s_b[2][LAMB_TILE];

    const uint32_t tid = threadIdx.x;
    const uint32_t slice = blockIdx.y;
    const uint32_t in0 = slice * slice_len;
    const uint32_t in1 = min(n_in, in0 + slice_len);
    const uint32_t n_tiles = (in1 - in0 + LAMB_TILE - 1) / LAMB_TILE;

    float nx[P], ny[P], nz[P];
    float ar[P], ag[P], ab[P];
    double dr[P], dg[P], db[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        uint32_t lo = (blockIdx.x * P + p) * LAMB_THREADS + tid;
        uint32_t o = out_begin + min(lo, out_count - 1);
        F3 n = ref_out_normal(o, out_size, out_size);
        nx[p] = n.x; ny[p] = n.y; nz[p] = n.z;
        ar[p] = ag[p] = ab[p] = 0.0f;
        dr[p] = dg[p] = db[p] = 0.0;
    }

    // One texel per thread per tile (LAMB_TILE == LAMB_THREADS).
    auto fetch = [&](uint32_t tile, float4& va, float2& vb) {
        uint32_t j = in0 + tile * LAMB_TILE + tid;
        if (j < in1) { va = a[j]; vb = b[j]; }
        else { va = make_float4(0.f, 0.f, 0.f, 0.f); vb = make_float2(0.f, 0.f); }
    };
    float4 va; float2 vb;
    fetch(0, va, vb);
    s_a[0][tid] = va; s_b[0][tid] = vb;
    __syncthreads();

    for (uint32_t tile = 0; tile < n_tiles; tile++) {
        const int cur = tile & 1;
        if (tile + 1 < n_tiles) fetch(tile + 1, va, vb);
#pragma unroll 4
        for (int i = 0; i < LAMB_TILE; i++) {
            const float4 ta = s_a[cur][i];
            const float2 tb = s_b[cur][i];
#pragma unroll
            for (int p = 0; p < P; p++) {
                float t = nx[p] * ta.x;
                t = fmaf(ny[p], ta.y, t);
                t = fma_clamp01(nz[p], ta.z, t);
                ar[p] = fmaf(t, ta.w, ar[p]);
                ag[p] = fmaf(t, tb.x, ag[p]);
                ab[p] = fmaf(t, tb.y, ab[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
            dr[p] += (double)ar[p]; dg[p] += (double)ag[p]; db[p] += (double)ab[p];
            ar[p] = ag[p] = ab[p] = 0.0f;
        }
        if (tile + 1 < n_tiles) { s_a[cur ^ 1][tid] = va; s_b[cur ^ 1][tid] = vb; }
        __syncthreads();
    }

#pragma unroll
    for (int p = 0; p < P; p++) {
        uint32_t lo = (blockIdx.x * P + p) * LAMB_THREADS + tid;
        if (lo < out_count) {
            size_t base = (size_t)slice * 3 * out_count + lo;
            partial[base] = dr[p];
            partial[base + out_count] = dg[p];
            partial[base + 2 * (size_t)out_count] = db[p];
        }
    }
}

__global__ __launch_bounds__(256) void lambertian_final(const double* __restrict__ partial, uint32_t n_slices,
                                                         uint32_t out_begin, uint32_t out_count, double scale,
                                                         float4* __restrict__ out) {
    uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    if (lo >= out_count) return;
    double r = 0.0, g = 0.0, b = 0.0;
    for (uint32_t s = 0; s < n_slices; s++) {
        size_t base = (size_t)s * 3 * out_count + lo;
        r += partial[base];
        g += partial[base + out_count];
        b += partial[base + 2 * (size_t)out_count];
    }
    out[out_begin + lo] = make_float4((float)(r * scale), (float)(g * scale), (float)(b * scale), 0.0f);
}

// ---- launch plan -----------------------------------------------------------

LambertianPlan lambertian_plan(uint32_t n_in, uint32_t out_count) {
    LambertianPlan pl;
    // Slice length depends on the input size only (sharding independence).
    uint32_t tiles = (n_in + LAMB_TILE - 1) / LAMB_TILE;
    uint32_t tiles_per_slice = (tiles + 127) / 128;          // at most 128 slices
    if (tiles_per_slice < 4) tiles_per_slice = 4;            // at least 1024 texels each
    pl.slice_len = tiles_per_slice * LAMB_TILE;
    pl.n_slices = (n_in + pl.slice_len - 1) / pl.slice_len;
    // Output texels per thread: as many (up to 4) as still leave >= ~1536 workgroups.
    int p = 4;   // 4 and 8 run at the same rate; 4 gives twice the workgroups
    if (int v = tuning().lamb_p) {                // tuning knob, not part of the ABI
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) p = v;
    } else
    while (p > 1) {
        uint64_t blocks = (uint64_t)((out_count + 256u * p - 1) / (256u * p)) * pl.n_slices;
        if (blocks >= 1536) break;
        p >>= 1;
    }
    pl.p = p;
    pl.out_blocks = (out_count + 256u * p - 1) / (256u * p);
    pl.partial_bytes = (size_t)pl.n_slices * 3 * out_count * sizeof(double);
    pl.prep_bytes_a = (size_t)n_in * sizeof(float4);
    pl.prep_bytes_b = (size_t)n_in * sizeof(float2);
    return pl;
}

hipError_t lambertian_launch(const LambertianPlan& pl, const float* d_env, uint32_t in_w, uint32_t in_h,
                             uint32_t out_size, uint32_t out_begin, uint32_t out_count, float* d_out,
                             void* d_a, void* d_b, double* d_partial, hipStream_t stream,
                             hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t n_in = in_w * in_h * 6;
    lambertian_prep<<<(n_in + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, in_h, n_in,
                                                            (float4*)d_a, (float2*)d_b);
    dim3 grid(pl.out_blocks, pl.n_slices);
    if (ev0) (void)hipEventRecord(ev0, stream);
    switch (pl.p) {
        case 16: lambertian_pairs<16><<<grid, LAMB_THREADS, 0, stream>>>((const float4*)d_a, (const float2*)d_b, n_in, pl.slice_len, out_size, out_begin, out_count, d_partial); break;
        case 8: lambertian_pairs<8><<<grid, LAMB_THREADS, 0, stream>>>((const float4*)d_a, (const float2*)d_b, n_in, pl.slice_len, out_size, out_begin, out_count, d_partial); break;
        case 4: lambertian_pairs<4><<<grid, LAMB_THREADS, 0, stream>>>((const float4*)d_a, (const float2*)d_b, n_in, pl.slice_len, out_size, out_begin, out_count, d_partial); break;
        case 2: lambertian_pairs<2><<<grid, LAMB_THREADS, 0, stream>>>((const float4*)d_a, (const float2*)d_b, n_in, pl.slice_len, out_size, out_begin, out_count, d_partial); break;
        default: lambertian_pairs<1><<<grid, LAMB_THREADS, 0, stream>>>((const float4*)d_a, (const float2*)d_b, n_in, pl.slice_len, out_size, out_begin, out_count, d_partial); break;
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
    const double scale = 4.0 * 0.31830988618379067154 / ((double)in_w * (double)in_h);
    lambertian_final<<<(out_count + 255) / 256, 256, 0, stream>>>(d_partial, pl.n_slices, out_begin, out_count, scale,
                                                                  (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
