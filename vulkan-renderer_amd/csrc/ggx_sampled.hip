// Sampled GGX prefilter (one mip level) for gfx950 -- an addition without a counterpart
// in the reference, whose integrateGGX (src/utilities/cube.cpp:152-309) visits every
// input texel.  This mode estimates the same normalised integral
//     sum L D(NoH) NoH NoL cw^2 / sum D(NoH) NoH NoL cw^2
// (NoL dw: cube.cpp:274-276; D NoH: :279-281; cutoff window cw: :284-290) with N half
// vectors drawn from D(h) NoH around n = v on Hammersley points, so a sample weighs
// NoH NoL cw^2, and reads the environment through a box pyramid at the level whose
// texel matches the sample's solid angle (1 / (N pdf)), trilinearly, with taps that
// leave a face taken from the neighbouring face.
//
// pyramid_level0 / pyramid_next   level k+1 = 2x2 box mean of level k (the tool's downscale
//     filter); every face of every level is stored with a one-texel apron that holds, for
//     each tap position just outside the face, the texel of the neighbouring face its
//     direction falls on -- so the sampling loop never leaves a face's own array.
// ggx_sampled_texels   one wavefront per output texel; lane j takes samples j, j+64, ...;
//     FP32 sums per lane, fixed-shape FP64 butterfly, so a texel's value depends on
//     (level, texel, N, bias) only.  The gathers are the work: 8 taps per sample.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include <math.h>

namespace ibl {

uint32_t pyramid_levels(uint32_t w) {
    uint32_t n = 1;
    while (w > 1 && (w & 1u) == 0) { w >>= 1; n++; }
    return n;
}

uint64_t pyramid_offset(uint32_t w, uint32_t level) {
    uint64_t off = 0;
    for (uint32_t k = 0; k < level; k++) off += (uint64_t)6 * ((w >> k) + 2) * ((w >> k) + 2);
    return off;
}

struct Pyramid {
    const float4* base;
    uint32_t w, levels;
    uint32_t off[16];   // float4 offset of each (padded) level; offsets fit 32 bits up to 8192-texel faces
};

// Per-sample arithmetic uses the 1-ulp hardware reciprocal, square root, logarithm and
// sine/cosine (the latter take revolutions, which is what the Hammersley coordinate is):
// a sample's direction and level move by ~1e-7, far below what the filtered lookup resolves.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Table of getNormalFrom (cube.cpp:35-46) for fractional (s, t).
__device__ __forceinline__ F3 face_dir(uint32_t face, float s, float t) {
    F3 r;
    switch (face) {
        case 0: r.x = 1.0f;  r.y = -t;    r.z = -s;   break;
        case 1: r.x = -1.0f; r.y = -t;    r.z = s;    break;
        case 2: r.x = s;     r.y = 1.0f;  r.z = t;    break;
        case 3: r.x = s;     r.y = -1.0f; r.z = -t;   break;
        case 4: r.x = s;     r.y = -t;    r.z = 1.0f; break;
        default: r.x = -s;   r.y = -t;    r.z = -1.0f; break;
    }
    return r;
}

// Its inverse: direction -> (face, s, t).  EXACT selects correctly rounded division: the
// rare taps that leave a face are snapped to a texel of the neighbouring face, and at small
// odd face sizes their direction can sit exactly on a texel boundary there, so that path
// repeats the CPU statement's arithmetic to the bit.
template <bool EXACT>
__device__ __forceinline__ void dir_to_face(F3 d, uint32_t& face, float& s, float& t) {
#pragma clang fp contract(off)
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) {
        const float inv = EXACT ? ref_div(1.0f, ax) : fast_rcp(ax);
        if (d.x > 0.0f) { face = 0; s = -d.z * inv; t = -d.y * inv; }
        else            { face = 1; s = d.z * inv;  t = -d.y * inv; }
    } else if (ay >= az) {
        const float inv = EXACT ? ref_div(1.0f, ay) : fast_rcp(ay);
        if (d.y > 0.0f) { face = 2; s = d.x * inv; t = d.z * inv; }
        else            { face = 3; s = d.x * inv; t = -d.z * inv; }
    } else {
        const float inv = EXACT ? ref_div(1.0f, az) : fast_rcp(az);
        if (d.z > 0.0f) { face = 4; s = d.x * inv;  t = -d.y * inv; }
        else            { face = 5; s = -d.x * inv; t = -d.y * inv; }
    }
}

// The texel a tap at (x, y) just outside face `face` of an S-texel level stands for: the
// one of the neighbouring face that the tap's direction falls on.  Correctly rounded
// arithmetic, the same as the CPU statement's: at small odd S such a direction can sit
// exactly on a texel boundary.
__device__ __forceinline__ void snap_tap(uint32_t& face, int& x, int& y, int S) {
#pragma clang fp contract(off)
    const float s = ref_div((float)x + 0.5f, (float)S) * 2.0f - 1.0f;
    const float t = ref_div((float)y + 0.5f, (float)S) * 2.0f - 1.0f;
    const F3 d = face_dir(face, s, t);
    float s2, t2;
    dir_to_face<true>(d, face, s2, t2);
    x = (int)floorf((s2 + 1.0f) * 0.5f * (float)S);
    y = (int)floorf((t2 + 1.0f) * 0.5f * (float)S);
    x = x < 0 ? 0 : (x >= S ? S - 1 : x);
    y = y < 0 ? 0 : (y >= S ? S - 1 : y);
}

// One thread per padded texel (x, y in [-1, S]) of level 0: copy, or the apron rule.
__global__ __launch_bounds__(256) void pyramid_level0(const float4* __restrict__ env, uint32_t S, float4* __restrict__ out) {
    const uint32_t P = S + 2, n = 6 * P * P;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t face = k / (P * P);
    const uint32_t rem = k - face * (P * P);
    int y = (int)(rem / P) - 1, x = (int)(rem % P) - 1;
    if (x < 0 || y < 0 || x >= (int)S || y >= (int)S) snap_tap(face, x, y, (int)S);
    out[k] = env[(face * S + (uint32_t)y) * S + (uint32_t)x];
}

// One thread per padded texel of the next level: the 2x2 box mean of the previous level's
// texels (sums in (row, column) order, times 1/4: downscale_box with factor 2) for the texel
// itself or for the texel its apron position stands for.
__global__ __launch_bounds__(256) void pyramid_next(const float4* __restrict__ prev, uint32_t S, float4* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t P = S + 2, n = 6 * P * P;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t face = k / (P * P);
    const uint32_t rem = k - face * (P * P);
    int y = (int)(rem / P) - 1, x = (int)(rem % P) - 1;
    if (x < 0 || y < 0 || x >= (int)S || y >= (int)S) snap_tap(face, x, y, (int)S);
    const uint32_t PP = 2 * S + 2;                       // padded edge of the previous level
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (uint32_t yy = 0; yy < 2; yy++)
        for (uint32_t xx = 0; xx < 2; xx++) {
            const float4 t = prev[(face * PP + (2 * (uint32_t)y + yy + 1)) * PP + (2 * (uint32_t)x + xx + 1)];
            ax += t.x; ay += t.y; az += t.z;
        }
    const float inv = ref_div(1.0f, 4.0f);
    out[k] = make_float4(inv * ax, inv * ay, inv * az, 0.0f);
}

hipError_t pyramid_launch(const float* d_env, uint32_t w, float* d_pyramid, hipStream_t stream) {
    const uint32_t levels = pyramid_levels(w);
    float4* base = (float4*)d_pyramid;
    {
        const uint32_t n = 6 * (w + 2) * (w + 2);
        pyramid_level0<<<(n + 255) / 256, 256, 0, stream>>>((const float4*)d_env, w, base);
    }
    for (uint32_t k = 1; k < levels; k++) {
        const uint32_t S = w >> k, n = 6 * (S + 2) * (S + 2);
        pyramid_next<<<(n + 255) / 256, 256, 0, stream>>>(base + pyramid_offset(w, k - 1), S, base + pyramid_offset(w, k));
    }
    return hipGetLastError();
}

// Texel (x, y), x and y in [-1, S], of a padded level.
__device__ __forceinline__ float4 pyr_texel(const Pyramid& p, uint32_t level, uint32_t face, int x, int y) {
    const uint32_t P = (p.w >> level) + 2;
    return p.base[p.off[level] + (face * P + (uint32_t)(y + 1)) * P + (uint32_t)(x + 1)];
}

__device__ __forceinline__ F3 pyr_bilinear(const Pyramid& p, uint32_t level, F3 dir) {
#pragma clang fp contract(off)
    uint32_t face; float s, t;
    dir_to_face<false>(dir, face, s, t);
    const float S = (float)(p.w >> level);
    const float fx = (s + 1.0f) * 0.5f * S - 0.5f, fy = (t + 1.0f) * 0.5f * S - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    // fx lies in [-0.5, S - 0.5], so the taps are x0, x0 + 1 in [-1, S]; the clamp only guards rounding
    const int smax = (int)S - 1;
    const int x0 = min(max((int)x0f, -1), smax), y0 = min(max((int)y0f, -1), smax);
    const float4 c00 = pyr_texel(p, level, face, x0, y0), c10 = pyr_texel(p, level, face, x0 + 1, y0);
    const float4 c01 = pyr_texel(p, level, face, x0, y0 + 1), c11 = pyr_texel(p, level, face, x0 + 1, y0 + 1);
    F3 r;
    { float a = c00.x + wx * (c10.x - c00.x), b = c01.x + wx * (c11.x - c01.x); r.x = a + wy * (b - a); }
    { float a = c00.y + wx * (c10.y - c00.y), b = c01.y + wx * (c11.y - c01.y); r.y = a + wy * (b - a); }
    { float a = c00.z + wx * (c10.z - c00.z), b = c01.z + wx * (c11.z - c01.z); r.z = a + wy * (b - a); }
    return r;
}

__device__ __forceinline__ F3 pyr_sample(const Pyramid& p, F3 dir, float lod) {
#pragma clang fp contract(off)
    const float top = (float)(p.levels - 1);
    lod = lod < 0.0f ? 0.0f : (lod > top ? top : lod);
    const float l0f = floorf(lod);
    const uint32_t l0 = (uint32_t)l0f;
    const float f = lod - l0f;
    F3 a = pyr_bilinear(p, l0, dir);
    if (f > 0.0f) {
        const F3 b = pyr_bilinear(p, l0 + 1, dir);
        a.x = a.x + f * (b.x - a.x); a.y = a.y + f * (b.y - a.y); a.z = a.z + f * (b.z - a.z);
    }
    return a;
}

__global__ __launch_bounds__(256) void ggx_sampled_texels(Pyramid p, uint32_t out_w, float a2, float cutoff,
                                                           uint32_t n_samples, float lod_bias, uint32_t out_begin,
                                                           uint32_t out_count, float4* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t local = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (local >= out_count) return;              // whole wave leaves: no barrier below
    const uint32_t o = out_begin + local;
    const F3 n = ref_out_normal(o, out_w, out_w);
    const F3 up = fabsf(n.z) < 0.999f ? F3{0.0f, 0.0f, 1.0f} : F3{1.0f, 0.0f, 0.0f};
    const F3 tx = ref_normalize(F3{up.y * n.z - up.z * n.y, up.z * n.x - up.x * n.z, up.x * n.y - up.y * n.x});
    const F3 bx = { n.y * tx.z - n.z * tx.y, n.z * tx.x - n.x * tx.z, n.x * tx.y - n.y * tx.x };
    const float inv_n = 1.0f / (float)n_samples;
    const float pi = 3.14159265358979323846f;
    const float inv_window = 1.0f / (1.0f - cutoff);
    const float lod_scale = inv_n * (4.0f * pi / a2) / (4.0f / ((float)p.w * (float)p.w));
    float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
    double dx = 0.0, dy = 0.0, dz = 0.0, dw = 0.0;
    uint32_t pending = 0;
    for (uint32_t i = lane; i < n_samples; i += 64) {
        const float u1 = ((float)i + 0.5f) * inv_n;
        const float u2 = (float)__brev(i) * 2.3283064365386963e-10f;
        const float cos2 = (1.0f - u2) * fast_rcp(1.0f + (a2 - 1.0f) * u2);
        const float nol = 2.0f * cos2 - 1.0f;
        if (!(nol > 0.0f) || !(cos2 > cutoff)) continue;
        const float cw = (cos2 - cutoff) * inv_window;
        const float cos_t = __builtin_amdgcn_sqrtf(cos2), sin_t = __builtin_amdgcn_sqrtf(1.0f - cos2);
        const float w = nol * cos_t * (cw * cw);
        const float cx = sin_t * __builtin_amdgcn_cosf(u1), cy = sin_t * __builtin_amdgcn_sinf(u1);   // cos, sin of 2 pi u1
        const F3 h = { cx * tx.x + cy * bx.x + cos_t * n.x, cx * tx.y + cy * bx.y + cos_t * n.y,
                       cx * tx.z + cy * bx.z + cos_t * n.z };
        const float k2 = 2.0f * cos_t;
        const F3 l = { k2 * h.x - n.x, k2 * h.y - n.y, k2 * h.z - n.z };
        const float d = cos2 * (a2 - 1.0f) + 1.0f;
        // solid angle of the sample over solid angle of a level-0 texel in that direction:
        // (1 / (N pdf)) / (4 m^3 / W^2), pdf = D / 4, D = a2 / (pi d^2)
        const float m = fmaxf(fabsf(l.x), fmaxf(fabsf(l.y), fabsf(l.z)));
        const float ratio = (d * d) * lod_scale * fast_rcp(m * m * m);
        const float lod = 0.5f * __builtin_amdgcn_logf(ratio) + lod_bias;
        const F3 c = pyr_sample(p, l, lod);
        ax += w * c.x; ay += w * c.y; az += w * c.z; aw += w;
        if (++pending == 64) {     // FP32 sums cover 64 samples, then go to FP64
            dx += (double)ax; dy += (double)ay; dz += (double)az; dw += (double)aw;
            ax = ay = az = aw = 0.0f; pending = 0;
        }
    }
    dx += (double)ax; dy += (double)ay; dz += (double)az; dw += (double)aw;
    for (int off = 32; off > 0; off >>= 1) {
        dx += __shfl_xor(dx, off); dy += __shfl_xor(dy, off); dz += __shfl_xor(dz, off); dw += __shfl_xor(dw, off);
    }
    if (lane == 0) {
        const double inv = 1.0 / dw;
        out[o] = make_float4((float)(inv * dx), (float)(inv * dy), (float)(inv * dz), 0.0f);
    }
}

hipError_t ggx_sampled_launch(const float* d_pyramid, uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips,
                              uint32_t n_samples, float lod_bias, uint32_t out_begin, uint32_t out_count, float* d_out,
                              hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const GgxLevel lv = ggx_level_params(mip, mips, in_w, in_w);
    Pyramid p;
    p.base = (const float4*)d_pyramid;
    p.w = in_w;
    p.levels = pyramid_levels(in_w);
    for (uint32_t k = 0; k < 16; k++) p.off[k] = k < p.levels ? (uint32_t)pyramid_offset(in_w, k) : 0u;
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (out_count)
        ggx_sampled_texels<<<(out_count + 3) / 4, 256, 0, stream>>>(p, out_w, lv.alpha2, lv.cutoff, n_samples, lod_bias,
                                                                    out_begin, out_count, (float4*)d_out);
    if (ev1) (void)hipEventRecord(ev1, stream);
    return hipGetLastError();
}

}  // namespace ibl
