// GGX specular prefilter (one mip level) for gfx950: the cutoff quadrature of
// integrateGGX, src/utilities/cube.cpp:152-309.
//
// For an output direction n and every input texel j (unit direction u_j, cube
// direction w_j, radiance L_j) inside a passing patch, with c = u_j.n and
// t = (1+c)/2 = cos^2 of the half-angle:
//     accept   t > cutoff  and  c > 0                       (cube.cpp:267-271)
//     weight   (c/|w_j|^3) * (a2*sqrt(t)/(pi*d^2)) * ((t-cutoff)/(1-cutoff))^2,
//              d = t*(a2-1)+1                               (cube.cpp:274-290)
//     out      sum(L_j * weight) / sum(weight)              (cube.cpp:292-305)
// The constants a2/pi and 1/(1-cutoff)^2 cancel in the quotient and are dropped.
// For small a2 the denominator d cancels to ~a2 at the lobe centre, so an error of
// one float ulp in t moves the weight by ~1e-7/a2.  The reference's t is
// NoH^2 = (|n|^2 + c)^2 / (|n|^2 + |u|^2 + 2c) with the float-normalised n and u it
// actually holds; to first order that is (1+c)/2 + 3/4 (|n|^2-1) - 1/4 (|u|^2-1).
// The |n|^2 term is the same for every input texel of one output texel (it would
// not average out), so it is folded into a per-output constant:
//     u = (1-c)/2 (exact in float for c >= 1/2),  d = u*(1-a2) + [a2 - (1-a2)*3/4*(|n|^2-1)].
// A patch is visited only when the reference's own patch test passes for that
// output texel (cube.cpp:256-258); that test is evaluated per lane in the
// reference's exact float arithmetic because it changes which texels count.
//
// Kernel plan
//   ggx_prep / ggx_prep_patches   per level: unit directions + |w|^-3 and
//       radiance*|w|^-3 per input texel, stored PATCH-MAJOR so the texels of one
//       patch are one contiguous run; unit centre direction per patch.
//   ggx_pairs   grid = (output tiles) x (interleaved patch slices).  A workgroup owns a tile
//       of 16PX x 16PY output texels (PX*PY per thread, directions in registers).
//       It walks its slice of patches 256 at a time: a conservative bounding-cone
//       test compacts the candidates into an LDS list in patch order, then each
//       candidate's texels are staged into LDS (double-buffered) and every lane
//       runs them against its own outputs, predicated by its exact patch test;
//       a wave with no passing lane skips the patch.  FP32 partial sums cover
//       one patch (<= 256 terms) and are then added to FP64 running sums.
//   ggx_final   sums the slices in order in FP64 and divides.
// Per-texel results do not depend on which sub-range of the level a call covers.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include <math.h>
#include <stdlib.h>

namespace ibl {

constexpr int GGX_THREADS = 256;
constexpr int GGX_PATCH_MAX = 256;   // 16 x 16 texels

uint32_t ggx_mip_levels(uint32_t top_w, uint32_t top_h, uint32_t min_size) {
    uint32_t m = top_w > top_h ? top_w : top_h;
    return (uint32_t)floor(log2((double)m) - log2((double)min_size));
}

float ggx_core_threshold(float alpha2) {
    // Measured against the CPU restatement: recomputing pairs with d < 2*a2 removes the
    // bias (mean error 1.5e-6 -> 1e-7 at roughness 0.2); wider cores and lobes with
    // a2 >= 0.005 gain nothing.  The multiplier is a diagnostic knob (Tuning::ggx_core).
    return alpha2 < 0.005f ? tuning().ggx_core * alpha2 : 0.0f;
}

GgxLevel ggx_level_params(uint32_t mip, uint32_t mips, uint32_t in_w, uint32_t in_h) {
    // Host floats, evaluated in the order of cube.cpp:161-170 (this file is
    // compiled without fused multiply-add on the host side: see build flags).
    GgxLevel lv;
    volatile float r = (float)mip / (float)mips;
    volatile float r2 = r * r;
    volatile float r3 = r2 * r;
    volatile float a2 = r3 * r;
    const float tol = 0.99f;
    volatile float num = a2 * tol;
    volatile float den0 = a2 - 1;
    volatile float den1 = den0 * tol;
    volatile float den = den1 + 1;
    volatile float q = num / den;
    volatile float c = 1 - q;
    lv.roughness = r;
    lv.alpha2 = a2;
    lv.cutoff = c > 0.95f ? 0.95f : c;
    uint32_t m = in_w < in_h ? in_w : in_h;
    lv.patch = m / 16 < 16u ? m / 16 : 16u;
    return lv;
}

__global__ __launch_bounds__(256) void ggx_prep(const float4* __restrict__ env, uint32_t w, uint32_t h, uint32_t ps,
                                                 uint32_t n_in, float4* __restrict__ a, float4* __restrict__ b) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_in) return;
    const uint32_t ps2 = ps * ps, pw = w / ps, ph = h / ps;
    uint32_t patch = k / ps2, sub = k - patch * ps2;
    uint32_t sy = sub / ps, sx = sub - sy * ps;
    uint32_t f = patch / (pw * ph);
    uint32_t rem = patch - f * (pw * ph);
    uint32_t py = rem / pw, px = rem - py * pw;
    uint32_t x = px * ps + sx, y = py * ps + sy;
    F3 d = ref_normal_from(f, x, y, w, h);
    F3 u = ref_normalize(d);
    float l2 = ref_dot(d, d);
    float s = 1.0f / (l2 * sqrtf(l2));
    float4 L = env[(f * h + y) * h + x];   // cube.cpp:264 row stride
    a[k] = make_float4(u.x, u.y, u.z, s);
    b[k] = make_float4(L.x * s, L.y * s, L.z * s, 0.0f);
}

__global__ __launch_bounds__(256) void ggx_prep_patches(uint32_t pw, uint32_t ph, uint32_t n_patches,
                                                         float4* __restrict__ pn) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_patches) return;
    uint32_t f = p / (pw * ph);
    uint32_t rem = p - f * (pw * ph);
    uint32_t y = rem / pw, x = rem - y * pw;
    F3 u = ref_normalize(ref_normal_from(f, x, y, pw, ph));   // cube.cpp:198
    pn[p] = make_float4(u.x, u.y, u.z, 0.0f);
}

template <int PX, int PY, bool COUNT, bool CORE>
__global__ __launch_bounds__(GGX_THREADS, 2) void ggx_pairs(
    const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ pn,
    uint32_t n_patches, uint32_t ps2,
    uint32_t out_w, uint32_t out_h, uint32_t tiles_x, uint32_t tiles_y,
    uint32_t out_begin, uint32_t out_count,
    float cutoff, float a2m1, float cmin, float theta_acc, float core_thr,
    double* __restrict__ partial, unsigned long long* __restrict__ counts) {
    constexpr int P = PX * PY;
    constexpr int GGX_UNROLL = CORE ? 1 : 2;   // the core branch defeats unrolling
    __shared__ float4 s_a[2][GGX_PATCH_MAX];
    __shared__ float4 s_b[2][GGX_PATCH_MAX];
    __shared__ uint32_t s_list[GGX_THREADS];
    __shared__ float s_red[4][GGX_THREADS / 64];
    __shared__ uint32_t s_wcnt[GGX_THREADS / 64];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t face = tile / (tiles_x * tiles_y);
    const uint32_t trem = tile - face * (tiles_x * tiles_y);
    const uint32_t tile_y = trem / tiles_x, tile_x = trem - tile_y * tiles_x;
    const uint32_t x0 = tile_x * (16 * PX) + (tid & 15) * PX;
    const uint32_t y0 = tile_y * (16 * PY) + (tid >> 4) * PY;

    float nx[P], ny[P], nz[P];
    float a2p[P];
    bool valid[P];
    uint32_t lo[P];
    const float om_a2 = -a2m1;   // 1 - a2
    float sumx = 0.f, sumy = 0.f, sumz = 0.f;
    bool any_valid = false;
#pragma unroll
    for (int p = 0; p < P; p++) {
        uint32_t x = x0 + (p % PX), y = y0 + (p / PX);
        bool in_face = x < out_w && y < out_h;
        uint32_t o = (face * out_h + min(y, out_h - 1)) * out_w + min(x, out_w - 1);
        valid[p] = in_face && o >= out_begin && (o - out_begin) < out_count;
        lo[p] = o - out_begin;
        F3 n = ref_normalize(ref_normal_from(face, min(x, out_w - 1), min(y, out_h - 1), out_w, out_h));
        nx[p] = n.x; ny[p] = n.y; nz[p] = n.z;
        {
            double eps = ((double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z) - 1.0;
            a2p[p] = (float)((1.0 + (double)a2m1) - (double)om_a2 * 0.75 * eps);
        }
        if (valid[p]) { sumx += n.x; sumy += n.y; sumz += n.z; any_valid = true; }
    }
    if (!__syncthreads_or(any_valid)) return;

    // Bounding cone of the tile's valid output directions.
    for (int off = 32; off > 0; off >>= 1) {
        sumx += __shfl_xor(sumx, off); sumy += __shfl_xor(sumy, off); sumz += __shfl_xor(sumz, off);
    }
    if (lane == 0) { s_red[0][wave] = sumx; s_red[1][wave] = sumy; s_red[2][wave] = sumz; }
    __syncthreads();
    float axx = 0.f, axy = 0.f, axz = 0.f;
    for (int w = 0; w < GGX_THREADS / 64; w++) { axx += s_red[0][w]; axy += s_red[1][w]; axz += s_red[2][w]; }
    {
        float inv = rsqrtf(axx * axx + axy * axy + axz * axz);
        axx *= inv; axy *= inv; axz *= inv;
    }
    float cmin_dot = 1.0f;
#pragma unroll
    for (int p = 0; p < P; p++)
        if (valid[p]) cmin_dot = fminf(cmin_dot, axx * nx[p] + axy * ny[p] + axz * nz[p]);
    for (int off = 32; off > 0; off >>= 1) cmin_dot = fminf(cmin_dot, __shfl_xor(cmin_dot, off));
    if (lane == 0) s_red[3][wave] = cmin_dot;
    __syncthreads();
    for (int w = 0; w < GGX_THREADS / 64; w++) cmin_dot = fminf(cmin_dot, s_red[3][w]);
    // A patch can pass some lane's test only within theta_acc of that lane's
    // direction, hence within theta_acc + cone radius of the axis (+ slack for
    // rounding in this float estimate).
    float block_thr;
    {
        float ang = theta_acc + acosf(fminf(fmaxf(cmin_dot, -1.0f), 1.0f)) + 4e-3f;
        block_thr = ang >= 3.14159f ? -2.0f : cosf(ang);
    }

    float aw[P], ar[P], ag[P], ab[P];
    double dw[P], dr[P], dg[P], db[P];
    float thr[P];
    uint32_t n_tested = 0, n_accepted = 0;
#pragma unroll
    for (int p = 0; p < P; p++) {
        aw[p] = ar[p] = ag[p] = ab[p] = 0.0f;
        dw[p] = dr[p] = dg[p] = db[p] = 0.0;
    }
    unsigned long long tot_tested = 0, tot_accepted = 0;

    // Slices interleave the patches (slice s owns q = s, s + S, s + 2S, ...) so that the
    // workgroups of one output tile carry equal shares of its cone.
    const uint32_t n_slices = gridDim.y;
    const uint32_t my_count = (n_patches + n_slices - 1 - blockIdx.y) / n_slices;

    for (uint32_t base = 0; base < my_count; base += GGX_THREADS) {
        // ---- candidate list of this chunk, in patch order --------------------
        const uint32_t qi = base + tid;
        uint32_t q = qi * n_slices + blockIdx.y;
        bool cand = false;
        if (qi < my_count) {
            float4 c4 = pn[q];
            cand = (c4.x * axx + c4.y * axy + c4.z * axz) >= block_thr;
        }
        unsigned long long bal = __ballot(cand);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        uint32_t offs = 0, n_cand = 0;
        for (uint32_t w = 0; w < GGX_THREADS / 64; w++) {
            uint32_t cw = s_wcnt[w];
            if (w < wave) offs += cw;
            n_cand += cw;
        }
        if (cand) s_list[offs + __popcll(bal & ((1ull << lane) - 1ull))] = q;
        __syncthreads();

        // ---- stream the candidates' texels through LDS ------------------------
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        if (n_cand > 0 && tid < ps2) {
            size_t k = (size_t)s_list[0] * ps2 + tid;
            va = a[k]; vb = b[k];
            s_a[0][tid] = va; s_b[0][tid] = vb;
        }
        __syncthreads();
        for (uint32_t ci = 0; ci < n_cand; ci++) {
            const int cur = ci & 1;
            const uint32_t patch = s_list[ci];
            if (ci + 1 < n_cand && tid < ps2) {
                size_t k = (size_t)s_list[ci + 1] * ps2 + tid;
                va = a[k]; vb = b[k];
            }
            // exact per-output patch test (cube.cpp:256-258)
            const float4 c4 = pn[patch];
            const F3 pc = { c4.x, c4.y, c4.z };
            bool lane_any = false;
#pragma unroll
            for (int p = 0; p < P; p++) {
                F3 n = { nx[p], ny[p], nz[p] };
                bool pass = valid[p] && ref_cone_test(pc, n, cutoff);
                thr[p] = pass ? cmin : 2.0f;
                lane_any |= pass;
                if (COUNT && pass) n_tested += ps2;
            }
            if (__any(lane_any)) {
#pragma unroll GGX_UNROLL
                for (uint32_t i = 0; i < ps2; i++) {
                    const float4 ta = s_a[cur][i];
                    const float4 tb = s_b[cur][i];
                    float cs[P], ds[P];
                    bool any_core = false;
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        float c = nx[p] * ta.x;
                        c = fmaf(ny[p], ta.y, c);
                        c = fmaf(nz[p], ta.z, c);
                        cs[p] = c;
                        ds[p] = fmaf(fmaf(-0.5f, c, 0.5f), om_a2, a2p[p]);
                        if (CORE) any_core |= ds[p] < core_thr;
                    }
                    // lobe core of a narrow lobe: the reference's own rounding of the
                    // cancelling denominator (see go_body in ggx_octets.hip)
                    if (CORE && __any(any_core)) {
                        const F3 uu = { ta.x, ta.y, ta.z };
#pragma unroll
                        for (int p = 0; p < P; p++) {
                            const F3 nn = { nx[p], ny[p], nz[p] };
                            const float den = ref_ggx_denominator(uu, nn, a2m1);
                            ds[p] = ds[p] < core_thr ? den : ds[p];
                        }
                    }
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const float c = cs[p];
                        float t = 1.0f - fmaf(-0.5f, c, 0.5f);
                        float sq = __builtin_amdgcn_sqrtf(t);
                        float qd = (t - cutoff) * __builtin_amdgcn_rcpf(ds[p]);
                        float f = (c * sq) * (qd * qd);
                        bool ok = c > thr[p];
                        f = ok ? f : 0.0f;
                        if (COUNT) n_accepted += ok ? 1u : 0u;
                        aw[p] = fmaf(f, ta.w, aw[p]);
                        ar[p] = fmaf(f, tb.x, ar[p]);
                        ag[p] = fmaf(f, tb.y, ag[p]);
                        ab[p] = fmaf(f, tb.z, ab[p]);
                    }
                }
#pragma unroll
                for (int p = 0; p < P; p++) {
                    dw[p] += (double)aw[p]; dr[p] += (double)ar[p];
                    dg[p] += (double)ag[p]; db[p] += (double)ab[p];
                    aw[p] = ar[p] = ag[p] = ab[p] = 0.0f;
                }
            }
            if (ci + 1 < n_cand && tid < ps2) { s_a[cur ^ 1][tid] = va; s_b[cur ^ 1][tid] = vb; }
            __syncthreads();
        }
        if (COUNT) { tot_tested += n_tested; tot_accepted += n_accepted; n_tested = 0; n_accepted = 0; }
    }

#pragma unroll
    for (int p = 0; p < P; p++) {
        if (valid[p]) {
            double4 v = make_double4(dw[p], dr[p], dg[p], db[p]);
            reinterpret_cast<double4*>(partial)[(size_t)blockIdx.y * out_count + lo[p]] = v;
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            tot_tested += __shfl_xor(tot_tested, off);
            tot_accepted += __shfl_xor(tot_accepted, off);
        }
        if (lane == 0) { atomicAdd(&counts[1], tot_tested); atomicAdd(&counts[2], tot_accepted); }
    }
}

__global__ __launch_bounds__(256) void ggx_final(const double* __restrict__ partial, uint32_t n_slices,
                                                  uint32_t out_begin, uint32_t out_count,
                                                  float4* __restrict__ out) {
    uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    if (lo >= out_count) return;
    double w = 0.0, r = 0.0, g = 0.0, b = 0.0;
    for (uint32_t s = 0; s < n_slices; s++) {
        double4 v = reinterpret_cast<const double4*>(partial)[(size_t)s * out_count + lo];
        w += v.x; r += v.y; g += v.z; b += v.w;
    }
    double inv = 1.0 / w;   // cube.cpp:305 via linear.hpp:93-95
    out[out_begin + lo] = make_float4((float)(inv * r), (float)(inv * g), (float)(inv * b), 0.0f);
}

// ---- launch plan -----------------------------------------------------------

GgxPlan ggx_plan(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h, uint32_t mip, uint32_t mips,
                 uint32_t out_count) {
    GgxPlan pl;
    pl.lv = ggx_level_params(mip, mips, in_w, in_h);
    const uint32_t ps = pl.lv.patch;
    pl.n_patches = 6 * (in_w / ps) * (in_h / ps);
    // Tile shape and slice count come from the level geometry only.
    const int shapes[3][2] = { {2, 2}, {2, 1}, {1, 1} };
    int pick = 2;
    for (int s = 0; s < 3; s++) {
        uint32_t tx = (out_w + 16 * shapes[s][0] - 1) / (16 * shapes[s][0]);
        uint32_t ty = (out_h + 16 * shapes[s][1] - 1) / (16 * shapes[s][1]);
        if (6ull * tx * ty >= 1024) { pick = s; break; }
    }
    pl.px = shapes[pick][0];
    pl.py = shapes[pick][1];
    pl.tiles_x = (out_w + 16 * pl.px - 1) / (16 * pl.px);
    pl.tiles_y = (out_h + 16 * pl.py - 1) / (16 * pl.py);
    uint32_t tiles = 6 * pl.tiles_x * pl.tiles_y;
    uint32_t want = (3072 + tiles - 1) / tiles;                 // slices wanted for ~3072 workgroups
    uint32_t max_slices = (pl.n_patches + 3) / 4;               // >= 4 patches per slice
    if (max_slices == 0) max_slices = 1;
    uint32_t slices = want < 1 ? 1 : (want > max_slices ? max_slices : want);
    pl.n_slices = slices;
    pl.partial_bytes = (size_t)pl.n_slices * out_count * 4 * sizeof(double);
    const size_t n_in = (size_t)pl.n_patches * ps * ps;
    pl.prep_bytes_a = n_in * sizeof(float4);
    pl.prep_bytes_b = n_in * sizeof(float4);
    pl.prep_bytes_pn = (size_t)pl.n_patches * sizeof(float4);
    return pl;
}

template <int PX, int PY>
static void ggx_dispatch(const GgxPlan& pl, dim3 grid, hipStream_t stream, const float4* a, const float4* b,
                         const float4* pn, uint32_t ps2, uint32_t out_w, uint32_t out_h, uint32_t out_begin,
                         uint32_t out_count, float a2m1, float cmin, float theta_acc, float core_thr, double* partial,
                         unsigned long long* counts) {
    auto go = [&](auto kernel) {
        kernel<<<grid, GGX_THREADS, 0, stream>>>(a, b, pn, pl.n_patches, ps2, out_w, out_h,
            pl.tiles_x, pl.tiles_y, out_begin, out_count, pl.lv.cutoff, a2m1, cmin, theta_acc, core_thr, partial, counts);
    };
    const bool core = core_thr > 0.0f;
    if (counts) { if (core) go(ggx_pairs<PX, PY, true, true>); else go(ggx_pairs<PX, PY, true, false>); }
    else        { if (core) go(ggx_pairs<PX, PY, false, true>); else go(ggx_pairs<PX, PY, false, false>); }
}

hipError_t ggx_launch(const GgxPlan& pl, const float* d_env, uint32_t in_w, uint32_t in_h,
                      uint32_t out_w, uint32_t out_h, uint32_t out_begin, uint32_t out_count, float* d_out,
                      void* d_a, void* d_b, void* d_pn, double* d_partial, unsigned long long* d_counts,
                      hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t ps = pl.lv.patch, ps2 = ps * ps;
    const uint32_t pw = in_w / ps, ph = in_h / ps;
    const uint32_t n_in = pl.n_patches * ps2;
    ggx_prep<<<(n_in + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, in_h, ps, n_in,
                                                     (float4*)d_a, (float4*)d_b);
    ggx_prep_patches<<<(pl.n_patches + 255) / 256, 256, 0, stream>>>(pw, ph, pl.n_patches, (float4*)d_pn);
    const double cut = (double)pl.lv.cutoff;
    const float cmin = (float)fmax(0.0, 2.0 * cut - 1.0);
    const float theta_acc = (float)acos(fmin(1.0, fmax(-1.0, 2.0 * cut - 1.0)));
    const float core_thr = ggx_core_threshold(pl.lv.alpha2);
    const float a2m1 = pl.lv.alpha2 - 1.0f;
    dim3 grid(6 * pl.tiles_x * pl.tiles_y, pl.n_slices);
    if (ev0) (void)hipEventRecord(ev0, stream);
    const float4* a = (const float4*)d_a; const float4* b = (const float4*)d_b; const float4* pn = (const float4*)d_pn;
    if (pl.px == 2 && pl.py == 2)
        ggx_dispatch<2, 2>(pl, grid, stream, a, b, pn, ps2, out_w, out_h, out_begin, out_count, a2m1, cmin, theta_acc, core_thr, d_partial, d_counts);
    else if (pl.px == 2)
        ggx_dispatch<2, 1>(pl, grid, stream, a, b, pn, ps2, out_w, out_h, out_begin, out_count, a2m1, cmin, theta_acc, core_thr, d_partial, d_counts);
    else
        ggx_dispatch<1, 1>(pl, grid, stream, a, b, pn, ps2, out_w, out_h, out_begin, out_count, a2m1, cmin, theta_acc, core_thr, d_partial, d_counts);
    if (ev1) (void)hipEventRecord(ev1, stream);
    ggx_final<<<(out_count + 255) / 256, 256, 0, stream>>>(d_partial, pl.n_slices, out_begin, out_count, (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
