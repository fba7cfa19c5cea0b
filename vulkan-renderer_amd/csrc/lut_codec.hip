// Split-sum BRDF table, RGBE codec and box downscale for gfx950.
//
// brdf_lut_cells   generateGGXLUT's cell loop, src/utilities/cube.cpp:346-385.
//     One wavefront per (roughness, cos theta_v) cell (four cells per 256-thread
//     workgroup); the samples_sqrt^2 strata are dealt round-robin to the 64
//     lanes, every lane keeps FP64 slope/bias sums, and a fixed-shape wavefront
//     butterfly combines them, so a cell's value depends on
//     (seed, cell, samples_sqrt) only.  No LDS.  The per-sample arithmetic keeps the
//     reference's mix of float and double where it matters (the cancelling
//     cos(theta) in double, BRDF terms in float); the sine and cosine of phi and
//     the fifth power run in float.  The reference seeds std::mt19937 from std::random_device; here
//     the jitter is a counter-based hash of (seed, cell, sample) or the stratum
//     midpoint.
// rgbe_decode / rgbe_encode   src/util.cpp:93-145, one texel per thread.
// texel_directions   getNormalFrom + normalize (cube.cpp:31-50, linear.hpp:114-122), one
//     texel per thread; the table both integrals start from.
// downscale_box    cube.cpp:482-501, one output texel per thread, float sums in
//     (row, column) order.
#include "ibl_device.h"
#include "ibl_kernels.h"

namespace ibl {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void brdf_lut_cells(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt,
                                                       uint64_t seed, int mode, uint32_t cell_begin,
                                                       uint32_t cell_count, float2* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t local = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (local >= cell_count) return;            // whole wave leaves: no barrier below
    const uint32_t cell = cell_begin + local;
    const uint32_t ri = cell / n_cos, ci = cell - ri * n_cos;
    const float roughness = (float)(((double)(float)ri + 0.5) / (double)n_rough);
    const float alpha = roughness * roughness;
    const float a2 = alpha * alpha;
    const float cos_v = (float)(((double)(float)ci + 0.5) / (double)n_cos);
    const float vx = (float)sqrt(1.0 - (double)(cos_v * cos_v));
    const float vz = cos_v;
    const float k = alpha / 2;
    const float g1v = vz / (vz * (1 - k) + k);
    const double inv_s = 1.0 / (double)samples_sqrt;
    const uint32_t n_samples = samples_sqrt * samples_sqrt;

    double slope = 0.0, bias = 0.0;
    for (uint32_t s = lane; s < n_samples; s += 64) {
        const uint32_t sx = s / samples_sqrt, sy = s - sx * samples_sqrt;
        double j1, j2;
        if (mode == 1) {
            uint64_t h = mix64(seed + 0x9e3779b97f4a7c15ull * ((((uint64_t)cell) << 32 | (uint64_t)s) + 1ull));
            j1 = (double)(uint32_t)(h & 0xffffffffu) * (1.0 / 4294967296.0) * inv_s;
            j2 = (double)(uint32_t)(h >> 32) * (1.0 / 4294967296.0) * inv_s;
        } else {
            j1 = 0.5 * inv_s; j2 = 0.5 * inv_s;
        }
        const float ux = (float)((double)((float)sx / (float)samples_sqrt) + j1);
        const float uy = (float)((double)((float)sy / (float)samples_sqrt) + j2);
        // cube.cpp:366-371 evaluates these in double and narrows each to float.  cos(theta) needs
        // that: 1 + (a2 - 1) uy cancels for small roughness.  sin(theta) is the square root of a
        // float-exact difference, and the sine and cosine of phi only have to be good to a few
        // ulp (their products with sin(theta) are narrowed to float anyway), so those run in
        // float: the double-precision sincos was most of the kernel's time.
        const float phi = (float)(2 * 3.14159265358979323846 * (double)ux);
        const float cos_t = (float)sqrt((1.0 - (double)uy) / (1.0 + ((double)a2 - 1.0) * (double)uy));
        const float sin_t = ref_sqrt(1.0f - cos_t * cos_t);
        float sphi, cphi;
        sincosf(phi, &sphi, &cphi);
        const float hx = sin_t * cphi;
        const float hy = sin_t * sphi;
        const float hz = cos_t;
        float vh = vx * hx;          // linear::dot with v.y == 0
        vh = vh + 0.0f * hy;
        vh = vh + vz * hz;
        const float two_vh = 2.0f * vh;
        const float lz = two_vh * hz - vz;
        if (lz > 0.0f) {
            const float g1l = lz / (lz * (1 - k) + k);
            const float g = g1l * g1v;
            const float omv = 1 - vh, omv2 = omv * omv;
            const float fres = omv2 * omv2 * omv;       // (1 - v.h)^5, cube.cpp:375
            const float norm = vh / (hz * vz);
            slope += (double)(g * (1 - fres) * norm);
            bias += (double)(g * fres * norm);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        slope += __shfl_xor(slope, off);
        bias += __shfl_xor(bias, off);
    }
    if (lane == 0) {
        slope /= (double)n_samples;
        bias /= (double)n_samples;
        out[cell] = make_float2((float)slope, (float)bias);
    }
}

hipError_t brdf_lut_launch(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int mode,
                           uint32_t cell_begin, uint32_t cell_count, float* d_out, hipStream_t stream,
                           hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (cell_count)
        brdf_lut_cells<<<(cell_count + 3) / 4, 256, 0, stream>>>(n_rough, n_cos, samples_sqrt, seed, mode, cell_begin,
                                                                 cell_count, (float2*)d_out);
    if (ev1) (void)hipEventRecord(ev1, stream);
    return hipGetLastError();
}

// ---- RGBE codec ---------------------------------------------------------------

__global__ __launch_bounds__(256) void rgbe_decode_k(const uchar4* __restrict__ src, float4* __restrict__ dst, uint32_t n) {
#pragma clang fp contract(off)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uchar4 s = src[i];
    float4 d = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (!(s.x == 0 && s.y == 0 && s.z == 0 && s.w == 0)) {
        int e = (int)s.w - 128;
        d.x = ldexpf(((float)s.x + 0.5f) / 256.0f, e);
        d.y = ldexpf(((float)s.y + 0.5f) / 256.0f, e);
        d.z = ldexpf(((float)s.z + 0.5f) / 256.0f, e);
    }
    dst[i] = d;
}

__device__ __forceinline__ unsigned char clamp_byte(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void rgbe_encode_k(const float4* __restrict__ src, uchar4* __restrict__ dst, uint32_t n) {
#pragma clang fp contract(off)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = src[i];
    float m = (s.x < s.y) ? s.y : s.x;
    m = (m < s.z) ? s.z : m;
    uchar4 d;
    if (m <= 1e-32f) {
        d = make_uchar4(0, 0, 0, 0);
    } else {
        int e;
        float frac_mult = ref_div(256.0f * frexpf(m, &e), m);
        if (e > 127) {
            d = make_uchar4(255, 255, 255, 255);
        } else {
            d.x = clamp_byte((int)(s.x * frac_mult));
            d.y = clamp_byte((int)(s.y * frac_mult));
            d.z = clamp_byte((int)(s.z * frac_mult));
            d.w = (unsigned char)(e + 128);
        }
    }
    dst[i] = d;
}

hipError_t rgbe_decode_launch(const uint8_t* d_rgbe, float* d_rgba, uint32_t n, hipStream_t stream) {
    if (n) rgbe_decode_k<<<(n + 255) / 256, 256, 0, stream>>>((const uchar4*)d_rgbe, (float4*)d_rgba, n);
    return hipGetLastError();
}

hipError_t rgbe_encode_launch(const float* d_rgba, uint8_t* d_rgbe, uint32_t n, hipStream_t stream) {
    if (n) rgbe_encode_k<<<(n + 255) / 256, 256, 0, stream>>>((const float4*)d_rgba, (uchar4*)d_rgbe, n);
    return hipGetLastError();
}

// ---- texel directions ------------------------------------------------------------

__global__ __launch_bounds__(256) void texel_directions(uint32_t w, uint32_t h, uint32_t n, int normalized,
                                                         float4* __restrict__ out) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const uint32_t f = o / (w * h), rem = o - f * (w * h);
    const uint32_t y = rem / w, x = rem - y * w;
    F3 d = ref_normal_from(f, x, y, w, h);
    if (normalized) d = ref_normalize(d);
    out[o] = make_float4(d.x, d.y, d.z, 0.0f);
}

hipError_t texel_directions_launch(uint32_t w, uint32_t h, int normalized, float* d_out, hipStream_t stream) {
    const uint32_t n = w * h * 6;
    if (n) texel_directions<<<(n + 255) / 256, 256, 0, stream>>>(w, h, n, normalized, (float4*)d_out);
    return hipGetLastError();
}

// ---- box downscale --------------------------------------------------------------

__global__ __launch_bounds__(256) void downscale_box(const float4* __restrict__ env, uint32_t w, uint32_t ow,
                                                      uint32_t n_out, uint32_t d, float4* __restrict__ out) {
#pragma clang fp contract(off)
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    uint32_t y = o / ow, x = o - y * ow;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (uint32_t yy = 0; yy < d; yy++) {
        const float4* row = env + (size_t)(y * d + yy) * w + (size_t)x * d;
        for (uint32_t xx = 0; xx < d; xx++) {
            float4 t = row[xx];
            ax += t.x; ay += t.y; az += t.z;
        }
    }
    const float inv = ref_div(1.0f, (float)(d * d));
    out[o] = make_float4(inv * ax, inv * ay, inv * az, 0.0f);
}

// The same filter for large factors.  One thread per output texel leaves a 2048 -> 32 filter with
// 6144 threads that each walk 4096 texels alone, far too little in flight to stream from HBM.  Here a
// workgroup owns `tpb` neighbouring output texels of one output row: all its threads copy the rows
// of those blocks (contiguous in memory) into LDS, a few rows at a time, and the first `tpb` threads
// then add their own block's texels from LDS -- still one float accumulator per texel and channel,
// in the reference's (row, column) order, so the bits do not change.  Blocks are padded to d + 1
// texels in LDS, which spreads the adders' reads over the banks.
constexpr uint32_t DS_THREADS = 256;
constexpr uint32_t DS_CHUNK = 1024;          // texels staged per round (16 KiB + padding)
#ifndef DS_TARGET_WGS
#define DS_TARGET_WGS 1536
#endif

__global__ __launch_bounds__(DS_THREADS) void downscale_box_staged(const float4* __restrict__ env, uint32_t w,
                                                                    uint32_t ow, uint32_t d, uint32_t log2d,
                                                                    uint32_t tpb, uint32_t rows_per_round,
                                                                    float4* __restrict__ out) {
#pragma clang fp contract(off)
    extern __shared__ float4 ds_tile[];
    const uint32_t tid = threadIdx.x;
    const uint32_t x0 = blockIdx.x * tpb;
    const uint32_t n_t = min(tpb, ow - x0);              // output texels of this workgroup
    const uint32_t oy = blockIdx.y;                      // output row within the 6-face strip
    const uint32_t span = n_t << log2d;                  // input texels per row, contiguous
    const float4* __restrict__ src = env + (size_t)oy * d * w + (size_t)x0 * d;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (uint32_t yy0 = 0; yy0 < d; yy0 += rows_per_round) {
        const uint32_t rows = min(rows_per_round, d - yy0);
        for (uint32_t r = 0; r < rows; r++) {
            const float4* __restrict__ row = src + (size_t)(yy0 + r) * w;
            for (uint32_t i = tid; i < span; i += DS_THREADS) {
                const uint32_t t = i >> log2d, xx = i & (d - 1);
                ds_tile[(r * tpb + t) * (d + 1) + xx] = row[i];
            }
        }
        __syncthreads();
        if (tid < n_t) {
            for (uint32_t r = 0; r < rows; r++) {
                const float4* __restrict__ mine = ds_tile + (r * tpb + tid) * (d + 1);
                for (uint32_t xx = 0; xx < d; xx += 8) {      // d is a multiple of 8 here
                    float4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = mine[xx + k];     // eight LDS reads in flight
#pragma unroll
                    for (int k = 0; k < 8; k++) { ax += v[k].x; ay += v[k].y; az += v[k].z; }
                }
            }
        }
        __syncthreads();
    }
    if (tid < n_t) {
        const float inv = ref_div(1.0f, (float)(d * d));
        out[(size_t)oy * ow + x0 + tid] = make_float4(inv * ax, inv * ay, inv * az, 0.0f);
    }
}

hipError_t downscale_launch(const float* d_env, uint32_t w, uint32_t h, uint32_t d, float* d_out, hipStream_t stream) {
    const uint32_t ow = w / d, oh = h / d;
    const uint32_t n_out = ow * oh * 6;
    if (n_out == 0) return hipGetLastError();
    if (d >= 8 && d <= DS_CHUNK && (d & (d - 1)) == 0) {    // the API only admits powers of two
        uint32_t log2d = 0;
        while ((1u << log2d) < d) log2d++;
        // Few output texels per workgroup, so that there are thousands of workgroups: each one's
        // adders walk their blocks alone, and only other workgroups can fill that time.
        uint32_t tpb = (n_out + DS_TARGET_WGS - 1) / DS_TARGET_WGS;
        if (tpb * d < 64) tpb = 64 / d;                  // at least 1 KiB contiguous per staged row
        if (tpb > 64) tpb = 64;
        if (tpb > ow) tpb = ow;
        if (tpb * d > DS_CHUNK) tpb = DS_CHUNK / d > 0 ? DS_CHUNK / d : 1;
        uint32_t rows = DS_CHUNK / (tpb * d);
        if (rows < 1) rows = 1;
        if (rows > d) rows = d;
        const size_t lds = (size_t)rows * tpb * (d + 1) * sizeof(float4);
        dim3 grid((ow + tpb - 1) / tpb, oh * 6);
        downscale_box_staged<<<grid, DS_THREADS, lds, stream>>>((const float4*)d_env, w, ow, d, log2d, tpb, rows,
                                                                (float4*)d_out);
    } else {
        downscale_box<<<(n_out + 255) / 256, 256, 0, stream>>>((const float4*)d_env, w, ow, n_out, d, (float4*)d_out);
    }
    return hipGetLastError();
}

}  // namespace ibl
