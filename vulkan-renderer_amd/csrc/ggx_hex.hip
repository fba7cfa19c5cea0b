// GGX specular prefilter (one mip level) through the 16-fold symmetry of the cube (gfx950).
//
// Same quadrature as ggx_octets.hip (integrateGGX, src/utilities/cube.cpp:152-309).  The texel-centre
// directions of an even, mirror-exact face grid are closed not only under the 8 sign flips but also under
// the exchange of x and y (face +-X <-> face +-Y, the Z faces transposed), and every float operation of the
// reference that involves two directions -- the dot products of linear.hpp:100-107 summed x, y, z, and
// with them the patch test (cube.cpp:256-258), the acceptance tests and the denominator (cube.cpp:267-281)
// -- gives the same bits when BOTH directions undergo the same flip or exchange: flips change no product,
// and the exchange swaps the first two terms of a sum that is formed (x + y) + z.  (The other axis
// permutations change the order of the additions and with it the rounding; they are not used.)
// Hence "hexadecets": groups of 16 texels, member m = (signs m & 7, exchange m >> 3) of canonical
// direction n being  D_signs R^exchange n.  For an output hexadecet with canonical n, an input one with
// canonical u and one of 16 patterns p = (tau, s), ONE cosine  nt . u,  nt = D_tau R^s n,  and ONE weight
// serve the 16 texel pairs
//     input member (sigma, 0)  ->  output member (sigma ^ tau, s)
//     input member (sigma, 1)  ->  output member (sigma ^ swapbits(tau), s ^ 1),
// twice as many as an octet's pattern serves: the weight arithmetic on the vector pipe, which bounds the
// octet kernel, is halved per texel pair, and the products on the matrix pipe (idle most of the time
// there) double per weight: 64 record columns instead of 32.
//
// Canonical hexadecets: every octet of the +-X faces (its partner is on the +-Y faces), and the octets
// (j, i) of the +-Z faces with i <= j (partner (i, j); on the diagonal the hexadecet has 8 distinct texels
// and only its first 8 members are stored).  Input patches pair up the same way; a patch on the diagonal
// of a Z face is its own partner, texel by texel transposed, so its records carry their first 8 members
// only (the other 8 columns and their weight column are zero) and every one of its texels is a record.
// Kernel plan: as ggx_octets.hip -- one wave per (unit of 64 output hexadecets = one 8 x 8 tile of
// canonical positions, run), bounding-cone candidates, the reference's exact patch test per lane and
// pattern, one walk over the patch's records per live pattern, 16 records at a time, the lanes evaluating
// the weights (the octet kernel's code) and v_mfma_f32_32x32x16_bf16 adding weight x column for two
// 32-column tiles; FP32 chunks of <= 256 records, the run's FP32 sums in LDS, the runs in FP64.
// Patches whose radiance sits in a few texels (ggx_hex_mark_patches, as ggx_oct_mark_patches) are left
// out by the main launch and walked by a second one (EXACT) with three bf16 pieces per factor and the six
// partial products that matter -- every FP32 x FP32 product to 2^-25 -- added into the same sums.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include "ggx_common.h"
#include <math.h>
#include <type_traits>

namespace ibl {

// Records per item of the launch that walks the marked patches: a whole patch.  (The octet kernel's vector
// walk waits out a scalar load per record when it is alone on its SIMD and is dealt 64 records at a time;
// what a wave of this kernel pays per item is mostly its set-up and the fold of 64 accumulators, so fewer,
// longer items: stretches of 64 records measured 1 % slower on the 512 chain.)
constexpr uint32_t GH_MARK_PART = 256;
constexpr int GH_SUMS = 50;                 // 16 members x RGB, then the weight sums of the two cosets
constexpr uint32_t GH_GROUP_WORDS = 6 * 64; // 16-byte words per group of 16 records: (upper, middle, lower piece) x 2 tiles x 64 lanes
constexpr uint32_t GH_REC_BYTES = 16 + GH_GROUP_WORDS * 16 / GO_MX_GROUP;   // direction + its share of a group's tiles

bool ggx_hex_eligible(uint32_t in_w, uint32_t out_w, uint32_t patch) {
    if (patch == 0 || in_w % patch) return false;
    const uint32_t pw = in_w / patch;
    // (from 32 x 32 output faces on: a 16 x 16 face is 2 units, too few waves -- 2048 -> 16 2.4 ms against the
    // octet kernel's 2.0, 512 -> 16 0.39 against 0.21; at 32 x 32 the two are level or this one ahead)
    return cube_is_mirror_exact(in_w) && cube_is_mirror_exact(out_w) && cube_is_mirror_exact(pw) &&
           out_w % (2 * GH_TILE) == 0 && out_w >= 32 && (patch * patch) % GO_MX_GROUP == 0;
}

uint32_t ggx_hex_units_total(uint32_t out_w) { return hex_units_total(out_w); }

// Does output unit u hold a texel whose (face, y, x) index lies in [begin, begin + count)?
__host__ __device__ inline bool hex_unit_in_range(uint32_t u, uint32_t out_w, uint32_t begin, uint32_t count) {
    for (uint32_t lane = 0; lane < GH_UNIT; lane++) {
        const HexLane h = hex_unit_lane(u, lane, out_w / 2);
        if (!h.valid) continue;
        for (uint32_t m = 0; m < (h.diagonal ? 8u : 16u); m++) {
            Texel t = hex_member(h.axis, h.j, h.i, m, out_w);
            if ((t.face * out_w + t.y) * out_w + t.x - begin < count) return true;
        }
    }
    return false;
}

uint32_t ggx_hex_units_in_range_count(uint32_t out_w, uint32_t begin, uint32_t count) {
    const uint32_t n_units = hex_units_total(out_w);
    uint32_t n = 0;
    for (uint32_t u = 0; u < n_units; u++) n += hex_unit_in_range(u, out_w, begin, count) ? 1u : 0u;
    return n;
}

// The list of those units in ascending order (one workgroup; the host knows its length from the count above).
__global__ __launch_bounds__(1024) void hex_units_in_range(uint32_t out_w, uint32_t n_units, uint32_t begin, uint32_t count,
                                                            uint32_t* __restrict__ list) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t u0 = 0; u0 < n_units; u0 += 1024) {
        const uint32_t u = u0 + tid;
        const bool hit = u < n_units && hex_unit_in_range(u, out_w, begin, count);
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_base, all = 0;
        for (uint32_t k = 0; k < 16; k++) { if (k < wave) before += s_wave[k]; all += s_wave[k]; }
        if (hit) list[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = u;
        __syncthreads();
        if (tid == 0) s_base += all;
        __syncthreads();
    }
}

hipError_t ggx_hex_units_in_range_launch(uint32_t out_w, uint32_t begin, uint32_t count, uint32_t* d_list, hipStream_t stream) {
    hex_units_in_range<<<1, 1024, 0, stream>>>(out_w, hex_units_total(out_w), begin, count, d_list);
    return hipGetLastError();
}

// ---- per-level tables ---------------------------------------------------------------------------------

// Canonical unit centre direction of every patch hexadecet (cube.cpp:198); .w: 1 where the patch is marked.
// Behind them one float4 whose .x holds the number of marked ones (as bits), then their list.
__global__ __launch_bounds__(256) void ggx_hex_prep_patches(uint32_t pw, uint32_t n_ph, float4* __restrict__ pn) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) pn[n_ph] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p >= n_ph) return;
    uint32_t axis, pj, pi;
    hex_patch_decode(p, pw / 2, axis, pj, pi);
    Texel c = octet_member(axis, pj, pi, 0, pw, pw);
    F3 u = ref_normalize(ref_normal_from(c.face, c.x, c.y, pw, pw));
    pn[p] = make_float4(u.x, u.y, u.z, 0.0f);
}

// Marks the patch hexadecets one of whose (up to 16) member patches holds its radiance in fewer than n_min
// equal texels, in some channel: S2 n_min > S1^2 with S1, S2 the sums of L and L^2 over the patch (see
// ggx_oct_mark_patches in ggx_octets.hip for why that bounds every output's error).  One workgroup per
// patch hexadecet, 32 threads per member and half.
__global__ __launch_bounds__(256) void ggx_hex_mark_patches(const float4* __restrict__ env, uint32_t w, uint32_t ps,
                                                             uint32_t n_ph, float n_min, float4* __restrict__ pn) {
    const uint32_t ph = blockIdx.x;
    const uint32_t l = threadIdx.x & 31u;
    const uint32_t ps2 = ps * ps;
    uint32_t axis, pj, pi;
    hex_patch_decode(ph, (w / ps) / 2, axis, pj, pi);
    const bool own_partner = axis == 2 && pj == pi;
    bool few = false;
    for (uint32_t m = threadIdx.x >> 5; m < (own_partner ? 8u : 16u); m += 8) {
        double s1[3] = {0.0, 0.0, 0.0}, s2[3] = {0.0, 0.0, 0.0};
        for (uint32_t sub = l; sub < ps2; sub += 32) {
            const uint32_t sy = sub / ps, sx = sub - sy * ps;
            Texel t = hex_member(axis, pj * ps + sy, pi * ps + sx, m, w);
            const float4 L = env[(t.face * w + t.y) * w + t.x];
            const double v[3] = { (double)L.x, (double)L.y, (double)L.z };
#pragma unroll
            for (int k = 0; k < 3; k++) { s1[k] += v[k]; s2[k] += v[k] * v[k]; }
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
            for (int off = 16; off > 0; off >>= 1) { s1[k] += __shfl_xor(s1[k], off); s2[k] += __shfl_xor(s2[k], off); }
#pragma unroll
        for (int k = 0; k < 3; k++) few = few || (s2[k] * (double)n_min > s1[k] * s1[k]);
    }
    if (__syncthreads_or(few ? 1 : 0) && threadIdx.x == 0) ((float*)(pn + ph))[3] = 1.0f;
}

// The marked patch hexadecets in ascending order behind the centres, and their number (one workgroup: the
// order decides which run a patch's terms are added in, so it must not depend on who got there first).
__global__ __launch_bounds__(1024) void ggx_hex_list_marks(uint32_t n_ph, float4* __restrict__ pn) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    uint32_t* __restrict__ list = (uint32_t*)(pn + n_ph + 1);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t b = 0; b < n_ph; b += 1024) {
        const uint32_t p = b + threadIdx.x;
        const bool m = p < n_ph && pn[p].w != 0.0f;
        const unsigned long long bal = __ballot(m);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_base, all = 0;
        for (uint32_t k = 0; k < 16; k++) { if (k < wv) before += s_wave[k]; all += s_wave[k]; }
        if (m) list[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
        if (threadIdx.x == 0) s_base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) ((uint32_t*)(pn + n_ph))[0] = s_base;
}

// The records of a level, patch-hexadecet-major: canonical unit direction and |w|^-3 of the canonical
// texel, four records to a 64-byte block (x0-3, y0-3, z0-3, |w|^-3 0-3: neighbouring records'
// components are scalar register pairs); then, per 16 consecutive records, six 1 KB tiles -- (upper,
// middle, lower bf16 piece) x (columns 0-31, columns 32-63) -- in the B-operand order of
// v_mfma_f32_32x32x16_bf16: lane l holds column l % 32 of records 8 (l / 32) + 0..7 as 8 bf16.
// Columns: 3 m + channel = radiance |w|^-3 of member m (m < 16); 48 = |w|^-3 for the first 8 members'
// weight sum, 49 = the same for the other 8 (0 on a patch that is its own partner, whose other 8 columns
// are 0 as well); 50-63 = 0.
__global__ __launch_bounds__(256) void ggx_hex_prep(const float4* __restrict__ env, uint32_t w, uint32_t ps,
                                                     uint32_t n_rec, float4* __restrict__ dirs, go_u32x4* __restrict__ tiles) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = t / 64, l = t % 64;
    if (grp * GO_MX_GROUP >= n_rec) return;
    const uint32_t half = l >> 5;
    const uint32_t ps2 = ps * ps, hpw = (w / ps) / 2;
    for (uint32_t tile = 0; tile < 2; tile++) {
        const uint32_t col = tile * 32 + (l & 31u);
        uint32_t pc[3][8];
#pragma unroll
        for (uint32_t e = 0; e < 8; e++) {
            const uint32_t k = grp * GO_MX_GROUP + half * 8 + e;
            const uint32_t ph = k / ps2, sub = k - ph * ps2;
            const uint32_t sy = sub / ps, sx = sub - sy * ps;
            uint32_t axis, pj, pi;
            hex_patch_decode(ph, hpw, axis, pj, pi);
            const bool own_partner = axis == 2 && pj == pi;
            const uint32_t j = pj * ps + sy, i = pi * ps + sx;
            Texel c = octet_member(axis, j, i, 0, w, w);
            F3 d = ref_normal_from(c.face, c.x, c.y, w, w);     // all components > 0
            float l2 = ref_dot(d, d);
            float s = 1.0f / (l2 * sqrtf(l2));
            float v = 0.0f;
            if (col < 48u) {
                const uint32_t m = col / 3u, ch = col - 3u * m;
                if (!(own_partner && m >= 8u)) {
                    Texel tx = hex_member(axis, j, i, m, w);
                    const float4 L = env[(tx.face * w + tx.y) * w + tx.x];
                    v = (ch == 0u ? L.x : ch == 1u ? L.y : L.z) * s;
                }
            } else if (col == 48u) {
                v = s;
            } else if (col == 49u) {
                v = own_partner ? 0.0f : s;
            }
            const uint32_t hi = go_bf16_bits(v);
            const float r1 = v - __uint_as_float(hi << 16);
            const uint32_t mid = go_bf16_bits(r1);
            const uint32_t lo = go_bf16_bits(r1 - __uint_as_float(mid << 16));
            pc[0][e] = hi; pc[1][e] = mid; pc[2][e] = lo;
            if (col == 0u) {      // four records to a 64-byte block: their x, their y, their z, their |w|^-3
                F3 u = ref_normalize(d);
                float* blk = (float*)(dirs + (k & ~3u));
                blk[k & 3u] = u.x; blk[4 + (k & 3u)] = u.y; blk[8 + (k & 3u)] = u.z; blk[12 + (k & 3u)] = s;
            }
        }
#pragma unroll
        for (uint32_t piece = 0; piece < 3; piece++) {
            go_u32x4 a;
#pragma unroll
            for (int q = 0; q < 4; q++) a[q] = pc[piece][2 * q] | (pc[piece][2 * q + 1] << 16);
            tiles[(size_t)grp * GH_GROUP_WORDS + (piece * 2 + tile) * 64 + l] = a;
        }
    }
}

// ---- the pair kernel ----------------------------------------------------------------------------------
#ifndef GH_WG_WAVES
#define GH_WG_WAVES 4           // most waves a workgroup may have (the launch bound)
#endif
#ifndef GH_MIN_WAVES
#define GH_MIN_WAVES 2          // (3: 168 registers; the plain one-group loop then has one spill in it and is 9 % slower on the 512
#endif                          //  and 2048 chains -- the SIMDs are busy at 2: vector and matrix instructions' times add up)
// EXACT: three bf16 pieces per factor and six partial products (marked patches).
// NARROW: the weight form for windows that end above the horizon (go_weight_quad).
// MARKS: 0 every patch hexadecet; 1 all but the marked ones; 2 only the marked ones, dealt to the runs as
// (patch, pattern, stretch of GH_MARK_PART records) items, numbered pattern-major, and the run's sums are ADDED to what the launch before
// (MARKS = 1, same grid) stored.
template <bool EXACT, bool NARROW, int MARKS>
__global__ __launch_bounds__(GO_WG * GH_WG_WAVES, EXACT ? 2 : GH_MIN_WAVES) void ggx_hex_pairs(
    const float4* __restrict__ rec, const float4* __restrict__ pn, const uint32_t* __restrict__ units,
    uint32_t n_ph, uint32_t ps2, uint32_t out_w, uint32_t oct_local,
    uint32_t runs_and_order, uint32_t out_begin, uint32_t out_count,
    float cutoff, float a2m1, float cmin, float theta_acc, float core_thr,
    float* __restrict__ partial, unsigned long long* __restrict__ counts) {
    static_assert(GO_WG == 64, "waves are independent: the grid below is counted in groups of GH_WG_WAVES of them");
    static_assert(EXACT == (MARKS == 2), "the exact products are for the marked patches");
    extern __shared__ float s_all[];       // per wave of the workgroup, the run's running sums: [sum][lane], rows padded
    float (*s_run)[GO_WG + 1] = (float (*)[GO_WG + 1])(s_all + (threadIdx.x >> 6) * (GH_SUMS * (GO_WG + 1)));
    const uint32_t wg_waves = blockDim.x >> 6;          // 1 ... GH_WG_WAVES
    const uint32_t lane = threadIdx.x & 63;
    // (run count of the level, first run of this launch and the grid order in one argument, as in ggx_oct_pairs)
    const uint32_t n_runs = runs_and_order & 0x7fffu;
    const uint32_t run_base = (runs_and_order >> 15) & 0x7fffu;
    const bool xcd_runs = (runs_and_order >> 31) != 0u;
    const uint32_t n_units = oct_local / GH_UNIT;
    // A workgroup is a few neighbouring units of one run: its waves share nothing but their compute unit,
    // where they walk the same patches at about the same time -- their direction blocks and column tiles
    // then come from the unit's scalar and vector caches instead of the L2 (the launch decides how many).
    const uint32_t n_ugroups = (n_units + wg_waves - 1) / wg_waves;
    uint32_t run_local, wv;
    if (xcd_runs) {                                                  // XCD k works on runs k, k + 8, ...
        const uint32_t seq = blockIdx.x / 8;
        run_local = blockIdx.x % 8 + 8 * (seq / n_ugroups);
        wv = (seq % n_ugroups) * wg_waves + (threadIdx.x >> 6);
    } else {
        run_local = blockIdx.x / n_ugroups;
        wv = (blockIdx.x % n_ugroups) * wg_waves + (threadIdx.x >> 6);
    }
    if (wv >= n_units) return;                                       // (no barrier anywhere: a wave may leave alone)
    const uint32_t run = run_base + run_local;
    const uint32_t mark_parts = (ps2 + GH_MARK_PART - 1) / GH_MARK_PART;
    // The bounding cones of the call's units (axis, cosine of the widest angle a candidate may have), behind
    // the list of marked patches: the launch that leaves the marked patches out stores them, the launch
    // that walks the marked ones reads them (cones the first did not store are not-a-number: nothing passes).
    float4* __restrict__ cones = (float4*)pn + (n_ph + 1u + (n_ph + 3u) / 4u);
    if (MARKS == 2) {
        // Before anything per lane: does one of this wave's items (marked patch, pattern, stretch of records)
        // lie inside its unit's cone at all?  With a handful of marked patches nearly every wave leaves here.
        const uint32_t per_pattern = mark_parts * __float_as_uint(pn[n_ph].x), n_items = 16u * per_pattern;
        if (run >= n_items) return;              // fewer items than runs: the sums of this run stand as they are
        const float4 cone = cones[wv];
        const uint32_t* __restrict__ list = (const uint32_t*)(pn + n_ph + 1);
        bool any = false;
        for (uint32_t it = run + lane * n_runs; it < n_items && !any; it += 64u * n_runs) {      // 64 items at a time, one per lane
            const uint32_t p = it / per_pattern;
            const float4 c4 = pn[list[(it - p * per_pattern) / mark_parts]];
            const float x = (p & 1u) ? -c4.x : c4.x, y = (p & 2u) ? -c4.y : c4.y, z = (p & 4u) ? -c4.z : c4.z;
            any = ((p & 8u) ? y : x) * cone.x + ((p & 8u) ? x : y) * cone.y + z * cone.z >= cone.w;
        }
        any = __any(any);
        if (!any) return;
    }
    const uint32_t lo = wv * GH_UNIT + lane;
    const HexLane hl = hex_unit_lane(units[wv], lane, out_w / 2);
    const bool lane_valid = hl.valid;

    // Members inside the requested (face, y, x) range decide whether the wave has anything to store, and
    // nothing else: a wave that runs adds for all its hexadecets, so that its sums do not depend on the range.
    {
        bool wanted = false;
        if (lane_valid) {
            for (uint32_t m = 0; m < (hl.diagonal ? 8u : 16u); m++) {
                Texel t = hex_member(hl.axis, hl.j, hl.i, m, out_w);
                const uint32_t o = (t.face * out_w + t.y) * out_w + t.x;
                wanted = wanted || (o - out_begin < out_count);
            }
        }
        if (!__any(wanted)) return;      // no barrier anywhere below: a wave may leave alone
    }

    Texel c0 = octet_member(hl.axis, hl.j, hl.i, 0, out_w, out_w);
    const F3 n = ref_normalize(ref_normal_from(c0.face, c0.x, c0.y, out_w, out_w));   // cube.cpp:242
    GoConst kc;
    kc.cutoff = cutoff; kc.om_a2 = -a2m1; kc.cmin = cmin; kc.core_thr = core_thr;
    float a2p;
    {
        double eps = ((double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z) - 1.0;
        a2p = (float)((1.0 + (double)a2m1) - (double)kc.om_a2 * 0.75 * eps);
    }

    // Bounding cone of the wave's canonical directions (see ggx_oct_pairs).
    float axx = lane_valid ? n.x : 0.f, axy = lane_valid ? n.y : 0.f, axz = lane_valid ? n.z : 0.f;
    for (int off = 32; off > 0; off >>= 1) {
        axx += __shfl_xor(axx, off); axy += __shfl_xor(axy, off); axz += __shfl_xor(axz, off);
    }
    {
        float inv = rsqrtf(axx * axx + axy * axy + axz * axz);
        axx *= inv; axy *= inv; axz *= inv;
    }
    float cmin_dot = lane_valid ? (axx * n.x + axy * n.y + axz * n.z) : 1.0f;
    for (int off = 32; off > 0; off >>= 1) cmin_dot = fminf(cmin_dot, __shfl_xor(cmin_dot, off));
    float wave_thr, core_cos;
    {
        const float radius = acosf(fminf(fmaxf(cmin_dot, -1.0f), 1.0f));
        float ang = theta_acc + radius + 4e-3f;
        wave_thr = ang >= 3.14159f ? -2.0f : cosf(ang);
        core_cos = 2.0f;                                         // no core: no patch qualifies
        if (core_thr > 0.0f) {
            const float u_core = (core_thr - (1.0f + a2m1)) / kc.om_a2 + 1e-5f;
            const float th_core = acosf(fminf(fmaxf(1.0f - 2.0f * u_core, -1.0f), 1.0f));
            // patches per face edge, from n_ph = 3/8 pw^2 + pw/4 (a little under, so the patch's radius a little over)
            const float pw = sqrtf((float)n_ph * (8.0f / 3.0f)) - 0.34f;
            ang = th_core + 1.4143f / pw + radius + 4e-3f;
            core_cos = ang >= 3.14159f ? -2.0f : cosf(ang);
        }
    }
    auto uniform = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    axx = uniform(axx); axy = uniform(axy); axz = uniform(axz);
    wave_thr = uniform(wave_thr); core_cos = uniform(core_cos);
    if (MARKS == 1 && run_local == 0u && lane == 0u) cones[wv] = make_float4(axx, axy, axz, wave_thr);

    // bf16 pairs (-1, -0) and (-0, -1) (see ggx_oct_pairs)
    constexpr uint32_t cut_lo = 0x8000bf80u, cut_hi = 0xbf808000u;
    // The chunk sums: four 32 x 32 accumulator tiles -- output hexadecets 0-31 and 32-63 of the wave x
    // record columns 0-31 and 32-63; this lane holds column lane % 32 of each, rows (i % 4) + 8 (i / 4) +
    // 4 (lane / 32) in element i.
    go_f32x16 mx[2][2];       // [column tile][half of the wave]
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 16; i++) mx[t][h][i] = 0.0f;
#pragma unroll
    for (int k = 0; k < GH_SUMS; k++) s_run[k][lane] = 0.0f;
    unsigned long long tot_walked = 0;        // wave-uniform

    // Column c of the tiles under pattern p belongs to sum hex_out_sum(c, p) of the output hexadecet.
    auto flush = [&](uint32_t p) __attribute__((always_inline)) {          // wave-uniform
        const uint32_t tau = p & 7u, s = p >> 3;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t col = 32u * t + (lane & 31u);
            uint32_t oc = 0xffffffffu;
            if (col < 48u) {
                const uint32_t m = col / 3u, ch = col - 3u * m;
                const uint32_t g = (m & 8u) ? (((m & 7u) ^ hex_swapbits(tau)) | ((s ^ 1u) << 3)) : (((m & 7u) ^ tau) | (s << 3));
                oc = g * 3u + ch;
            } else if (col == 48u) {
                oc = 48u + s;
            } else if (col == 49u) {
                oc = 48u + (s ^ 1u);
            }
            if (oc != 0xffffffffu) {
                float* q = &s_run[oc][4u * (lane >> 5)];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    // (read, add, write: the LDS's own add, ds_add_f32, needs no wait but runs at a fraction of
                    // the rate -- the 2048 chain took 2.11 s with it instead of 1.17)
                    const int row = (i & 3) + 8 * (i >> 2);
                    q[row] += mx[t][0][i];
                    q[32 + row] += mx[t][1][i];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) { mx[t][0][i] = 0.0f; mx[t][1][i] = 0.0f; }
        }
    };
    uint32_t acc_p = 0, acc_terms = 0;        // wave-uniform: the pattern and the size of the open chunk
    uint32_t walked_any = 0;                  // wave-uniform (MARKS == 2)
    uint32_t w_begin = 0, w_count = ps2;      // wave-uniform: the records a walk covers (a stretch with MARKS == 2)

    // One candidate patch hexadecet: pmask holds the patterns whose transformed patch centre lies inside
    // the wave's cone (bits 0-15) and, of those, the ones close enough for the lobe core (bits 16-31).
    auto visit = [&](uint32_t ph, uint32_t pmask) __attribute__((always_inline)) {
        // exact per-lane patch tests (cube.cpp:256-258), one per candidate pattern: canonical n against the
        // patch centre with the pattern's flips and exchange applied
        const float4 c4 = pn[ph];
        uint32_t lane_pass = 0;
#pragma unroll
        for (int p = 0; p < 16; p++) {
            if (pmask & (1u << p)) {    // wave-uniform
                const float x = (p & 1) ? -c4.x : c4.x, y = (p & 2) ? -c4.y : c4.y, z = (p & 4) ? -c4.z : c4.z;
                const F3 pc = (p & 8) ? F3{ y, x, z } : F3{ x, y, z };
                if (lane_valid && ref_cone_test(pc, n, cutoff)) lane_pass |= 1u << p;
            }
        }
        uint32_t wmask_v = 0;
#pragma unroll
        for (int p = 0; p < 16; p++)
            if (__any((lane_pass >> p) & 1u)) wmask_v |= 1u << p;
        uint32_t wmask = __builtin_amdgcn_readfirstlane(wmask_v);    // the same in every lane
        if (MARKS == 2) walked_any |= wmask;
        // ---- one walk over the patch's records per live pattern
        while (wmask) {
            const uint32_t p = (uint32_t)__builtin_ctz(wmask);
            wmask &= wmask - 1;
            if (acc_terms != 0u && (acc_p != p || acc_terms + w_count > (uint32_t)GO_CHUNK)) { flush(acc_p); acc_terms = 0u; }
            acc_p = p;
            acc_terms += w_count;
            tot_walked += w_count;
            // this lane's cutoff for the walk (2 if its patch test failed: weight 0)
            const float thr = ((lane_pass >> p) & 1u) ? cutoff : 2.0f;
            // nt = D_tau R^s n
            const float sx = (p & 8u) ? n.y : n.x, sy = (p & 8u) ? n.x : n.y;
            const F3 nt = { __uint_as_float(__float_as_uint(sx) ^ ((p & 1u) << 31)),
                            __uint_as_float(__float_as_uint(sy) ^ ((p & 2u) << 30)),
                            __uint_as_float(__float_as_uint(n.z) ^ ((p & 4u) << 29)) };
            const bool core_near = ((pmask >> (16u + p)) & 1u) != 0u;       // wave-uniform
            auto walk = [&](auto core_c) __attribute__((always_inline)) {
                constexpr bool CORE = decltype(core_c)::value;
                const float4* __restrict__ dp = rec + (size_t)ph * ps2 + w_begin;                 // wave-uniform
                const go_u32x4* __restrict__ tp = (const go_u32x4*)(rec + (size_t)n_ph * ps2)
                                                  + ((size_t)ph * ps2 + w_begin) / GO_MX_GROUP * GH_GROUP_WORDS + lane;
                GoDir4 da, db;
                // four records: weights, cut into their bf16 pieces, two records to a register
                auto quad = [&](const GoDir4& src, uint32_t& h0, uint32_t& h1, uint32_t& m0, uint32_t& m1, uint32_t& l0, uint32_t& l1)
                                __attribute__((always_inline)) {
                    float f[4];
                    if constexpr (!CORE) {
                        auto f4 = [&](int o) __attribute__((always_inline)) {
                            const go_f32x4 v = { __uint_as_float(src.v[o]), __uint_as_float(src.v[o + 1]),
                                                 __uint_as_float(src.v[o + 2]), __uint_as_float(src.v[o + 3]) };
                            return v;
                        };
                        const go_f32x4 nx = { nt.x, nt.x, nt.x, nt.x }, ny = { nt.y, nt.y, nt.y, nt.y },
                                       nz = { nt.z, nt.z, nt.z, nt.z };
                        go_f32x4 c = nx * f4(0);
                        c = __builtin_elementwise_fma(ny, f4(4), c);
                        c = __builtin_elementwise_fma(nz, f4(8), c);
                        const go_f32x4 w = go_weight_quad<NARROW>(c, thr, a2p, kc);
                        f[0] = w.x; f[1] = w.y; f[2] = w.z; f[3] = w.w;
                    } else {
                        uint32_t none = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float ux = __uint_as_float(src.v[e]), uy = __uint_as_float(src.v[4 + e]),
                                        uz = __uint_as_float(src.v[8 + e]);
                            float c = nt.x * ux;
                            c = fmaf(nt.y, uy, c);
                            c = fmaf(nt.z, uz, c);
                            f[e] = go_weight<false, true>(c, ux, uy, uz, thr, a2p, kc, nt, none);
                            asm volatile("" : "+v"(f[e]));      // (one record at a time: see ggx_oct_pairs)
                        }
                    }
                    h0 = go_pack_bf16(f[0], f[1]);
                    h1 = go_pack_bf16(f[2], f[3]);
                    if constexpr (EXACT) {
                        const float r0 = f[0] - __uint_as_float(h0 << 16), r1 = f[1] - __uint_as_float(h0 & 0xffff0000u);
                        const float r2 = f[2] - __uint_as_float(h1 << 16), r3 = f[3] - __uint_as_float(h1 & 0xffff0000u);
                        m0 = go_pack_bf16(r0, r1);
                        m1 = go_pack_bf16(r2, r3);
                        l0 = go_pack_bf16(r0 - __uint_as_float(m0 << 16), r1 - __uint_as_float(m0 & 0xffff0000u));
                        l1 = go_pack_bf16(r2 - __uint_as_float(m1 << 16), r3 - __uint_as_float(m1 & 0xffff0000u));
                    } else if constexpr (!CORE) {
                        // rounded weight minus its upper piece in one instruction, h . (-1, -0) + f
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        auto sub = [](uint32_t h, uint32_t k, float x) {
                            return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, k), x, false);
                        };
                        m0 = go_pack_bf16(sub(h0, cut_lo, f[0]), sub(h0, cut_hi, f[1]));
                        m1 = go_pack_bf16(sub(h1, cut_lo, f[2]), sub(h1, cut_hi, f[3]));
                    } else {
                        m0 = go_pack_bf16(f[0] - __uint_as_float(h0 << 16), f[1] - __uint_as_float(h0 & 0xffff0000u));
                        m1 = go_pack_bf16(f[2] - __uint_as_float(h1 << 16), f[3] - __uint_as_float(h1 & 0xffff0000u));
                    }
                };
                // A group of 16 records: the A operands are their weights' pieces (rows = output hexadecets, so the
                // upper lanes' records 0-7 and the lower lanes' records 8-15 change places), the B operands the
                // group's record columns, two tiles.
                struct Pieces { go_u32x4 a0h, a1h, a0m, a1m, a0l, a1l; };
                struct Columns { go_u32x4 hi[2], mid[2], lo[2]; };
                auto column = [&](Columns& c, const go_u32x4* __restrict__ tg, int t) __attribute__((always_inline)) {
                    c.hi[t] = tg[t * 64];
                    c.mid[t] = tg[128 + t * 64];
                    if constexpr (EXACT) c.lo[t] = tg[256 + t * 64];
                };
                // quarter q of a group's matrix instructions (small terms first)
                auto products = [&](const Pieces& pc, const Columns& c, int q) __attribute__((always_inline)) {
                    auto bf = [](const go_u32x4& v) { return __builtin_bit_cast(go_bf16x8, v); };
                    auto mm = [&](const go_u32x4& a0, const go_u32x4& a1, const go_u32x4& bb, int t) __attribute__((always_inline)) {
                        mx[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0), bf(bb), mx[t][0], 0, 0, 0);
                        mx[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(bb), mx[t][1], 0, 0, 0);
                    };
                    if constexpr (EXACT) {
                        // 24 instructions: quarter q takes tile q / 2, and of its six terms the first or the last three
                        const int t = q / 2;
                        if (q % 2 == 0) { mm(pc.a0l, pc.a1l, c.hi[t], t); mm(pc.a0h, pc.a1h, c.lo[t], t); mm(pc.a0m, pc.a1m, c.mid[t], t); }
                        else { mm(pc.a0m, pc.a1m, c.hi[t], t); mm(pc.a0h, pc.a1h, c.mid[t], t); mm(pc.a0h, pc.a1h, c.hi[t], t); }
                    } else {
                        // 12 instructions, 3 per quarter: (term, tile) pairs in the order m.hi, h.mid, h.hi per tile
                        switch (q) {
                            case 0: mm(pc.a0m, pc.a1m, c.hi[0], 0); mx[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(pc.a0h), bf(c.mid[0]), mx[0][0], 0, 0, 0); break;
                            case 1: mx[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(pc.a1h), bf(c.mid[0]), mx[0][1], 0, 0, 0); mm(pc.a0h, pc.a1h, c.hi[0], 0); break;
                            case 2: mm(pc.a0m, pc.a1m, c.hi[1], 1); mx[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(pc.a0h), bf(c.mid[1]), mx[1][0], 0, 0, 0); break;
                            default: mx[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(pc.a1h), bf(c.mid[1]), mx[1][1], 0, 0, 0); mm(pc.a0h, pc.a1h, c.hi[1], 1); break;
                        }
                    }
                };
                // The weights of group g, and between them -- four records at a time -- the matrix instructions of the
                // group before (WITH): a wave issues in order, and a group's matrix instructions in a row at its end
                // keep the wave off the vector port for 384 cycles that the one other wave of the SIMD cannot fill.
                // (The statements with the scalar loads and their waits are fixed points of the compiler's schedule,
                // so the place of the matrix instructions in the source is their place in the code.)
                auto weights = [&](uint32_t g, auto with_c, const Pieces& before, const go_u32x4* __restrict__ tg) __attribute__((always_inline)) {
                    constexpr bool WITH = decltype(with_c)::value;
                    uint32_t ah[8], am[8], al[8];
                    Columns cols;
                    if constexpr (WITH) column(cols, tg, 0);      // (the second tile's columns are fetched halfway: 8 registers)
                    go_dir_wait(da);
                    go_dir_load(db, dp + g + 4);
                    quad(da, ah[0], ah[1], am[0], am[1], al[0], al[1]);
                    if constexpr (WITH) products(before, cols, 0);
                    go_dir_wait(db);
                    go_dir_load(da, dp + g + 8);
                    quad(db, ah[2], ah[3], am[2], am[3], al[2], al[3]);
                    if constexpr (WITH) { column(cols, tg, 1); products(before, cols, 1); }
                    go_dir_wait(da);
                    go_dir_load(db, dp + g + 12);
                    quad(da, ah[4], ah[5], am[4], am[5], al[4], al[5]);
                    if constexpr (WITH) products(before, cols, 2);
                    go_dir_wait(db);
                    go_dir_load(da, dp + min(g + 16, w_count - 4));
                    quad(db, ah[6], ah[7], am[6], am[7], al[6], al[7]);
                    if constexpr (WITH) products(before, cols, 3);
                    Pieces pc;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        auto sh = __builtin_amdgcn_permlane32_swap(ah[i], ah[4 + i], false, false);
                        pc.a0h[i] = sh[0]; pc.a1h[i] = sh[1];
                        auto sm = __builtin_amdgcn_permlane32_swap(am[i], am[4 + i], false, false);
                        pc.a0m[i] = sm[0]; pc.a1m[i] = sm[1];
                        if constexpr (EXACT) {
                            auto sl = __builtin_amdgcn_permlane32_swap(al[i], al[4 + i], false, false);
                            pc.a0l[i] = sl[0]; pc.a1l[i] = sl[1];
                        }
                    }
                    return pc;
                };
                go_dir_load(da, dp);
                // (two groups per turn, the pieces in two sets that change roles: one set and a copy per group
                // is 15 moves per group, and the compiler declines to unroll a loop with the scalar loads in it)
                auto finish = [&](const Pieces& last) __attribute__((always_inline)) {
                    Columns cols;
                    column(cols, tp, 0);
                    column(cols, tp, 1);
#pragma unroll
                    for (int q = 0; q < 4; q++) products(last, cols, q);
                };
                if constexpr (EXACT) {
                    // (twice the matrix instructions and a third set of pieces: no room for two groups at once, and
                    // the matrix pipe is the bound here anyway)
                    for (uint32_t g = 0; g < w_count; g += GO_MX_GROUP) {
                        const Pieces pc = weights(g, std::false_type{}, Pieces{}, tp);
                        finish(pc);
                        tp += GH_GROUP_WORDS;
                    }
                    go_dir_wait(da);
                    return;
                }
                Pieces pa = weights(0, std::false_type{}, Pieces{}, tp);
                uint32_t g = GO_MX_GROUP;
                for (; g + GO_MX_GROUP < w_count; g += 2 * GO_MX_GROUP) {
                    const Pieces pb = weights(g, std::true_type{}, pa, tp);
                    pa = weights(g + GO_MX_GROUP, std::true_type{}, pb, tp + GH_GROUP_WORDS);
                    tp += 2 * GH_GROUP_WORDS;
                }
                if (g < w_count) {          // an even number of groups: one more before the last one's products
                    const Pieces pb = weights(g, std::true_type{}, pa, tp);
                    tp += GH_GROUP_WORDS;
                    finish(pb);
                } else {
                    finish(pa);
                }
                go_dir_wait(da);       // nothing of this walk is in flight when it ends
            };
            if (core_near) walk(std::true_type{});
            else walk(std::false_type{});
        }
    };

    if constexpr (MARKS == 2) {
        // This run's items -- run, run + n_runs, ... -- 64 at a time, one per lane, as the other launches scan patches.
        // Items are numbered pattern-major (all stretches of all marked patches under pattern 0, then under
        // pattern 1, ...): a unit's cone mostly holds patches under one pattern, and with the pattern in the low
        // bits of the number those would all fall to runs of one residue modulo 8 -- one XCD of the eight.
        const uint32_t* __restrict__ list = (const uint32_t*)(pn + n_ph + 1);
        const uint32_t per_pattern = mark_parts * __float_as_uint(pn[n_ph].x), n_items = 16u * per_pattern;
        for (uint32_t first = run; first < n_items; first += 64u * n_runs) {
            const uint32_t it = first + lane * n_runs;
            uint32_t ph_v = 0, code = 0;          // code: pattern | stretch << 4 | near the core << 31 | candidate << 30
            if (it < n_items) {
                const uint32_t p = it / per_pattern, within = it - p * per_pattern;
                const uint32_t part = within % mark_parts;
                ph_v = list[within / mark_parts];
                const float4 c4 = pn[ph_v];
                const float x = (p & 1u) ? -c4.x : c4.x, y = (p & 2u) ? -c4.y : c4.y, z = (p & 4u) ? -c4.z : c4.z;
                const float dsum = ((p & 8u) ? y : x) * axx + ((p & 8u) ? x : y) * axy + z * axz;
                code = p | (part << 4) | (dsum >= core_cos ? 0x80000000u : 0u) | (dsum >= wave_thr ? 0x40000000u : 0u);
            }
            unsigned long long bal = __ballot((code & 0x40000000u) != 0u);
            while (bal) {                                  // ascending item order
                const int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                const uint32_t ph = __builtin_amdgcn_readlane(ph_v, src), cd = __builtin_amdgcn_readlane(code, src);
                const uint32_t p = cd & 15u;
                w_begin = ((cd >> 4) & 0xffffu) * GH_MARK_PART;
                w_count = min(ps2 - w_begin, GH_MARK_PART);
                visit(ph, (1u << p) | ((cd >> 31) ? (65536u << p) : 0u));
            }
        }
    } else {
        // The level's patch hexadecets are dealt round-robin to the runs (run r owns q = r, r + n_runs, ...).
        const uint32_t my_count = (n_ph + n_runs - 1 - run) / n_runs;
        for (uint32_t base = 0; base < my_count; base += 64) {
            const uint32_t qi = base + lane;
            const uint32_t q = qi * n_runs + run;
            uint32_t cmask = 0;
            if (qi < my_count) {
                const float4 c4 = pn[q];
                // without the exchange: x with x, y with y; with it: the centre's y against the axis' x
                const float a = c4.x * axx, b = c4.y * axy, c = c4.z * axz;
                const float a1 = c4.y * axx, b1 = c4.x * axy;
#pragma unroll
                for (int tau = 0; tau < 8; tau++) {
                    const float d0 = ((tau & 1) ? -a : a) + ((tau & 2) ? -b : b) + ((tau & 4) ? -c : c);
                    const float d1 = ((tau & 2) ? -a1 : a1) + ((tau & 1) ? -b1 : b1) + ((tau & 4) ? -c : c);
                    if (d0 >= wave_thr) cmask |= 1u << tau;
                    if (d0 >= core_cos) cmask |= 65536u << tau;
                    if (d1 >= wave_thr) cmask |= 256u << tau;
                    if (d1 >= core_cos) cmask |= 16777216u << tau;
                }
                if (MARKS == 1 && c4.w != 0.0f) cmask = 0;
            }
            unsigned long long bal = __ballot((cmask & 65535u) != 0u);
            while (bal) {                                  // ascending patch order
                const int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                visit(__builtin_amdgcn_readlane(q, src), __builtin_amdgcn_readlane(cmask, src));
            }
        }
    }
    if (acc_terms != 0u) flush(acc_p);

    if (MARKS == 2) {
        if (walked_any != 0u && lane_valid) {
#pragma unroll
            for (int k = 0; k < GH_SUMS; k++)
                partial[((size_t)run_local * GH_SUMS + k) * oct_local + lo] += s_run[k][lane];
        }
    } else if (lane_valid) {
#pragma unroll
        for (int k = 0; k < GH_SUMS; k++)
            partial[((size_t)run_local * GH_SUMS + k) * oct_local + lo] = s_run[k][lane];
    }
    // records this wave walked (64 lanes each, once per live pattern): what the arithmetic executed is priced by
    if (counts && lane == 0) atomicAdd(&counts[3], tot_walked);
}

// First stage of the run sum when a level has many runs (as ggx_oct_reduce): one thread per (output
// hexadecet, group of sums, group of runs); sums 3 m .. 3 m + 2 for m < 16, the two weight sums for m = 16.
__global__ __launch_bounds__(256) void ggx_hex_reduce(const float* __restrict__ partial, uint32_t n_runs,
                                                       uint32_t per_group, uint32_t group_base, uint32_t oct_count,
                                                       uint32_t add_to, double* __restrict__ reduced) {
    const uint32_t lo = blockIdx.x * GO_THREADS + threadIdx.x;
    if (lo >= oct_count) return;
    const uint32_t m = blockIdx.y, g = blockIdx.z;
    const uint32_t s0 = g * per_group, s1 = min(n_runs, s0 + per_group);
    const uint32_t j0 = m * 3, nj = m < 16 ? 3 : 2;
    double v[3] = {0.0, 0.0, 0.0};
    if (add_to)
        for (uint32_t k = 0; k < nj; k++) v[k] = reduced[((size_t)(group_base + g) * GH_SUMS + (j0 + k)) * oct_count + lo];
    for (uint32_t sl = s0; sl < s1; sl++)
        for (uint32_t k = 0; k < nj; k++) v[k] += (double)partial[((size_t)sl * GH_SUMS + (j0 + k)) * oct_count + lo];
    for (uint32_t k = 0; k < nj; k++) reduced[((size_t)(group_base + g) * GH_SUMS + (j0 + k)) * oct_count + lo] = v[k];
}

// One thread per (output hexadecet, member): the runs (or the groups of runs) in order, the quotient of
// cube.cpp:305, the texel stored if the range asks for it.
template <typename T>
__global__ __launch_bounds__(256) void ggx_hex_final(const T* __restrict__ partial, uint32_t n_sums, uint32_t out_w,
                                                      const uint32_t* __restrict__ units, uint32_t oct_count,
                                                      uint32_t out_begin, uint32_t out_count, float4* __restrict__ out) {
    const uint32_t lo = blockIdx.x * GO_THREADS + threadIdx.x;
    if (lo >= oct_count) return;
    const uint32_t m = blockIdx.y;
    const HexLane hl = hex_unit_lane(units[lo / GH_UNIT], lo % GH_UNIT, out_w / 2);
    if (!hl.valid || (hl.diagonal && m >= 8u)) return;
    Texel t = hex_member(hl.axis, hl.j, hl.i, m, out_w);
    const uint32_t o = (t.face * out_w + t.y) * out_w + t.x;
    if (o - out_begin >= out_count) return;
    double v[3] = {0.0, 0.0, 0.0}, w = 0.0;
    for (uint32_t sl = 0; sl < n_sums; sl++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += (double)partial[((size_t)sl * GH_SUMS + (m * 3 + k)) * oct_count + lo];
        w += (double)partial[((size_t)sl * GH_SUMS + 48 + (m >> 3)) * oct_count + lo];
    }
    double inv = 1.0 / w;   // cube.cpp:305 via linear.hpp:93-95
    out[o] = make_float4((float)(inv * v[0]), (float)(inv * v[1]), (float)(inv * v[2]), 0.0f);
}

// ---- launch plan ----------------------------------------------------------------------------------------

GgxOctPlan ggx_hex_plan(uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips, uint32_t n_units) {
    GgxOctPlan pl;
    pl.lv = ggx_level_params(mip, mips, in_w, in_w);
    const uint32_t ps = pl.lv.patch, pw = in_w / ps;
    pl.hex = true;
    pl.matrix = true;
    pl.n_po = hex_patches_total(pw);
    pl.tw = GH_TILE;
    const uint32_t units_level = hex_units_total(out_w);
    // (at most 200 000 waves: a wave of this kernel sets up more than an octet wave does, and with the 4064 runs the
    // shared rule gives the 2048 -> 128 level a run is one or two patches: 51.2 ms, 48.7 with 1000 or 2000 runs)
    ggx_plan_runs(pl, units_level * GH_UNIT, units_level, 16, GH_SUMS, (size_t)GH_REC_BYTES, n_units, 200000.0);
    pl.marks = tuning().ggx_few > 0.0f;
    pl.mx_bytes = (size_t)pl.n_po * ps * ps * (size_t)GH_REC_BYTES;
    pl.rec_bytes = pl.mx_bytes;
    // centres, the number of marked patches, their list (padded to whole float4), the units' cones
    pl.pn_bytes = ((size_t)pl.n_po + 1 + (pl.n_po + 3) / 4 + n_units) * sizeof(float4);
    return pl;
}

hipError_t ggx_hex_launch(const GgxOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_w,
                          const uint32_t* d_units, uint32_t out_begin, uint32_t out_count,
                          float* d_out, void* d_rec, void* d_pn, double* d_partial, unsigned long long* d_counts,
                          hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t ps = pl.lv.patch, ps2 = ps * ps, pw = in_w / ps;
    const uint32_t n_rec = pl.n_po * ps2;
    ggx_hex_prep_patches<<<(pl.n_po + 255) / 256, 256, 0, stream>>>(pw, pl.n_po, (float4*)d_pn);
    if (pl.marks) {
        const float n_min = tuning().ggx_few > 256.0f ? tuning().ggx_few : fminf(tuning().ggx_few, 0.5f * (float)ps2);
        ggx_hex_mark_patches<<<pl.n_po, 256, 0, stream>>>((const float4*)d_env, in_w, ps, pl.n_po, n_min, (float4*)d_pn);
        ggx_hex_list_marks<<<1, 1024, 0, stream>>>(pl.n_po, (float4*)d_pn);
        // (all bits set: cones that no wave stores -- units outside the call's range -- let nothing pass)
        if (hipError_t e = hipMemsetAsync((float4*)d_pn + (pl.n_po + 1 + (pl.n_po + 3) / 4), 0xff,
                                          (size_t)pl.out_units * sizeof(float4), stream); e != hipSuccess) return e;
    }
    ggx_hex_prep<<<(n_rec / GO_MX_GROUP * 64 + 255) / 256, 256, 0, stream>>>(
        (const float4*)d_env, in_w, ps, n_rec, (float4*)d_rec, (go_u32x4*)((float4*)d_rec + n_rec));
    const double cut = (double)pl.lv.cutoff;
    const float cmin = (float)fmax(0.0, 2.0 * cut - 1.0);
    const float theta_acc = (float)acos(fmin(1.0, fmax(-1.0, 2.0 * cut - 1.0)));
    const float a2m1 = pl.lv.alpha2 - 1.0f;
    const float core_thr = ggx_core_threshold(pl.lv.alpha2);
    // (grid order: the rule of ggx_oct_launch)
    const double cone = 0.5 * (1.0 - fmax(0.0, 2.0 * (double)pl.lv.cutoff - 1.0));
    const double run_bytes = (double)pl.mx_bytes / (double)pl.n_runs;
    uint32_t xcd_runs = (pl.runs_per_pass % 8 == 0 &&
                         (pl.out_units < 3072u || 16.0 * cone < 3.0 || run_bytes > (double)(2u << 20))) ? 1u : 0u;
    if (tuning().ggx_xcd_slices >= 0) xcd_runs = (tuning().ggx_xcd_slices != 0 && pl.runs_per_pass % 8 == 0) ? 1u : 0u;
    // (waves per workgroup: 4 neighbouring units from 1024^2 inputs on -- 2 % on the 2048 chain, nothing or a
    // little less than nothing on the 512 chain, 8 is 8 % slower on both)
    uint32_t wg_waves = in_w >= 1024u ? GH_WG_WAVES : 1u;
    if (tuning().ggx_wg_waves >= 1) wg_waves = (uint32_t)tuning().ggx_wg_waves < (uint32_t)GH_WG_WAVES ? (uint32_t)tuning().ggx_wg_waves : (uint32_t)GH_WG_WAVES;
    const dim3 grid(pl.runs_per_pass * ((pl.out_units + wg_waves - 1) / wg_waves));
    const size_t lds_bytes = (size_t)wg_waves * GH_SUMS * (GO_WG + 1) * sizeof(float);
    const uint32_t oct_count = pl.out_units * GH_UNIT;
    const uint32_t small_blocks = (oct_count + GO_THREADS - 1) / GO_THREADS;
    double* d_groups = d_partial;
    const bool two_stage = pl.accumulate || pl.reduce_groups > 1;
    float* d_runs = (float*)(d_partial + (size_t)(two_stage ? pl.reduce_groups : 0) * GH_SUMS * oct_count);
    // wide lobes: patches further than 90 degrees + their radius from every direction of a wave have weight 0
    // throughout (cube.cpp:271), so the wave's cone test leaves them out (see ggx_oct_launch)
    const bool wide = cmin == 0.0f && tuning().ggx_early != 0;
    const float theta_walk = wide ? fminf(theta_acc, 1.5707964f + 1.4143f / (float)pw) : theta_acc;
    const bool narrow = pl.lv.cutoff >= 0.51f && (tuning().ggx_rsq == 2 || (tuning().ggx_rsq == 1 && in_w >= 1024u));
    if (ev0) (void)hipEventRecord(ev0, stream);
    for (uint32_t pass = 0; pass < pl.passes; pass++) {
        const uint32_t run_base = pass * pl.runs_per_pass;
        const uint32_t packed = pl.n_runs | (run_base << 15) | (xcd_runs << 31);
        auto go = [&](auto kernel) {
            kernel<<<grid, GO_WG * wg_waves, lds_bytes, stream>>>((const float4*)d_rec, (const float4*)d_pn, d_units, pl.n_po, ps2, out_w,
                oct_count, packed, out_begin, out_count, pl.lv.cutoff, a2m1, cmin, theta_walk, core_thr, d_runs, d_counts);
        };
        if (pl.marks) {
            if (narrow) go(ggx_hex_pairs<false, true, 1>); else go(ggx_hex_pairs<false, false, 1>);
            if (narrow) go(ggx_hex_pairs<true, true, 2>); else go(ggx_hex_pairs<true, false, 2>);
        } else {
            if (narrow) go(ggx_hex_pairs<false, true, 0>); else go(ggx_hex_pairs<false, false, 0>);
        }
        if (pass + 1 == pl.passes && ev1) (void)hipEventRecord(ev1, stream);
        if (pl.accumulate)
            ggx_hex_reduce<<<dim3(small_blocks, 17, 1), GO_THREADS, 0, stream>>>(
                d_runs, pl.runs_per_pass, pl.runs_per_pass, 0, oct_count, pass > 0 ? 1u : 0u, d_groups);
        else if (pl.reduce_groups > 1)
            ggx_hex_reduce<<<dim3(small_blocks, 17, pl.passes > 1 ? pl.runs_per_pass / pl.per_group : pl.reduce_groups),
                             GO_THREADS, 0, stream>>>(d_runs, pl.runs_per_pass, pl.per_group, run_base / pl.per_group,
                                                      oct_count, 0u, d_groups);
    }
    if (two_stage)
        ggx_hex_final<double><<<dim3(small_blocks, 16), GO_THREADS, 0, stream>>>(
            d_groups, pl.reduce_groups, out_w, d_units, oct_count, out_begin, out_count, (float4*)d_out);
    else
        ggx_hex_final<float><<<dim3(small_blocks, 16), GO_THREADS, 0, stream>>>(
            d_runs, pl.n_runs, out_w, d_units, oct_count, out_begin, out_count, (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
