// GGX specular prefilter (one mip level) through the cube's mirror symmetry (gfx950).
//
// Same quadrature as ggx.hip (integrateGGX, src/utilities/cube.cpp:152-309), evaluated
// octet against octet.  Output texels, input texels and input patches all come in groups
// of 8 mirror images (sign flips of x, y, z).  For an output octet with canonical
// (all-positive) unit direction n and an input texel octet with canonical unit direction
// u, the cosine between output member so and input member si depends only on the
// component-wise sign product t = so (x) si:
//     c_t = tx*nx*ux + ty*ny*uy + tz*nz*uz.
// The GGX weight is a function of that cosine (times |w|^-3, which all 8 input members
// share), and so is the reference's patch test (cube.cpp:256-258).  Hence, per (output
// octet, input texel octet) and per sign pattern t that passes the patch test:
// one cosine, ONE weight evaluation, then 8 x 4 FMAs that add it to the sums of the
// 8 output members, each paired with input member so^t -- about 48 VALU instructions
// for 8 texel pairs where the general kernel spends about 21 per pair.  The reference
// computes each of those 8 cosines from the same operands in the same order, so
// sharing them changes nothing it could see.
//
// Kernel plan
//   ggx_oct_prep / ggx_oct_prep_patches   per level: one 7-float4 record per input texel
//       octet (canonical unit direction, |w|^-3, radiance*|w|^-3 of the 8 members), stored
//       patch-octet-major; canonical unit centre direction per patch octet.
//   ggx_oct_pairs   grid = (units of 64 output octets in tile-major order) x (interleaved
//       slices of patch octets), one wave per workgroup; every wave works alone, and a
//       level is cut into at least 131072 of them.  Per 64 patch octets a
//       bounding-cone test marks, per lane, the candidate (patch octet, sign patterns); the
//       wave visits the candidates in patch order (ballot + readlane); every lane runs the
//       reference's exact float patch test for its own octet and each candidate pattern; the
//       wave walks the patch's records once, reading them with wave-uniform (scalar) loads,
//       and evaluates the patterns that pass for any of its lanes.  FP32 sums cover 64
//       records, then go to FP64 running sums.  No LDS, no barriers.
//   ggx_oct_reduce / ggx_oct_final   per output octet: slices summed in a fixed two-stage
//       order in FP64, divided, scattered to the member texels that lie in the requested
//       (face, y, x) range.
// A texel's value does not depend on the range or shard a call covers.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include <math.h>
#include <stdlib.h>

namespace ibl {

constexpr int GO_THREADS = 256;         // threads per workgroup of the small kernels
constexpr int GO_UNIT = 64;             // output octets per wave: the unit of sharding
#ifndef GO_WG
#define GO_WG 64                        // threads per workgroup of ggx_oct_pairs (waves are independent)
#endif
constexpr int GO_REC = 7;              // float4 per texel-octet record
constexpr int GO_SUMS = 25;             // doubles per (slice, output octet): 8 members x RGB + the weight sum
constexpr int GO_FLUSH = 64;           // records per FP32 chunk when all 8 patterns are live

// Output octets are numbered tile-major: tiles of tw x tw canonical positions.
__host__ __device__ inline uint32_t go_tile_edge(uint32_t hw) {
    uint32_t tw = hw < 16 ? hw : 16;
    while (hw % tw) tw--;
    return tw;
}

__device__ __forceinline__ void go_decode(uint32_t g, uint32_t hw, uint32_t tw, uint32_t& axis, uint32_t& oj, uint32_t& oi) {
    const uint32_t t2 = tw * tw, tpr = hw / tw;
    const uint32_t tile = g / t2, within = g - tile * t2;
    const uint32_t ly = within / tw, lx = within - ly * tw;
    axis = tile / (tpr * tpr);
    const uint32_t trem = tile - axis * (tpr * tpr);
    const uint32_t ty = trem / tpr, tx = trem - ty * tpr;
    oj = ty * tw + ly;
    oi = tx * tw + lx;
}

// A call owns the 64-octet units first_unit, first_unit + unit_stride, ... of the
// tile-major octet order; lo indexes its own octets (and its partial sums).
__device__ __forceinline__ uint32_t go_global(uint32_t lo, uint32_t first_unit, uint32_t unit_stride) {
    return ((lo / GO_UNIT) * unit_stride + first_unit) * GO_UNIT + (lo % GO_UNIT);
}

__global__ __launch_bounds__(256) void ggx_oct_prep(const float4* __restrict__ env, uint32_t w, uint32_t ps,
                                                     uint32_t n_rec, float4* __restrict__ rec) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rec) return;
    const uint32_t ps2 = ps * ps, hpw = (w / ps) / 2;
    const uint32_t po = k / ps2, sub = k - po * ps2;
    const uint32_t sy = sub / ps, sx = sub - sy * ps;
    const uint32_t axis = po / (hpw * hpw);
    const uint32_t prem = po - axis * (hpw * hpw);
    const uint32_t pj = prem / hpw, pi = prem - pj * hpw;
    const uint32_t j = pj * ps + sy, i = pi * ps + sx;
    Texel c = octet_member(axis, j, i, 0, w, w);
    F3 d = ref_normal_from(c.face, c.x, c.y, w, w);     // all components > 0
    F3 u = ref_normalize(d);
    float l2 = ref_dot(d, d);
    float s = 1.0f / (l2 * sqrtf(l2));
    float v[28];
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = s;
#pragma unroll
    for (uint32_t m = 0; m < 8; m++) {
        Texel t = octet_member(axis, j, i, m, w, w);
        float4 L = env[(t.face * w + t.y) * w + t.x];
        v[4 + 3 * m] = L.x * s; v[5 + 3 * m] = L.y * s; v[6 + 3 * m] = L.z * s;
    }
#pragma unroll
    for (int q = 0; q < GO_REC; q++)
        rec[(size_t)k * GO_REC + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

__global__ __launch_bounds__(256) void ggx_oct_prep_patches(uint32_t pw, uint32_t n_po, float4* __restrict__ pn) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_po) return;
    const uint32_t hpw = pw / 2;
    const uint32_t axis = p / (hpw * hpw);
    const uint32_t rem = p - axis * (hpw * hpw);
    const uint32_t pj = rem / hpw, pi = rem - pj * hpw;
    Texel c = octet_member(axis, pj, pi, 0, pw, pw);
    F3 u = ref_normalize(ref_normal_from(c.face, c.x, c.y, pw, pw));   // cube.cpp:198
    pn[p] = make_float4(u.x, u.y, u.z, 0.0f);
}

struct GoConst {
    float cutoff, om_a2, cmin;
    float core_thr;   // below this denominator the reference's own rounding is reproduced (0: never)
};

// One sign pattern TAU of one record: weight from the cosine, added to all 8 members.
template <int TAU, bool COUNT>
__device__ __forceinline__ void go_body(float c, const float (&r)[28], float thr, float a2p, const GoConst& k,
                                        const F3& n, float (&acc)[8][4], uint32_t& n_acc) {
    float u = fmaf(-0.5f, c, 0.5f);
    float t = 1.0f - u;
    float sq = __builtin_amdgcn_sqrtf(t);
    float d = fmaf(u, k.om_a2, a2p);
    // At the centre of a narrow lobe d cancels to ~a2, the weight ~1/d^2 amplifies the
    // reference's float rounding of NoH^2 (about 1e-7/a2 per pair), and that rounding is
    // not zero-mean there (1/sqrt of |half|^2 just below 4 lands on rounding ties).  Inside
    // the lobe core the denominator is therefore recomputed with the reference's exact
    // operation sequence.  Output tiles are small against the core, so the branch is
    // nearly wave-uniform.
    const bool core = d < k.core_thr;
    if (__any(core)) {
        F3 us = { (TAU & 1) ? -r[0] : r[0], (TAU & 2) ? -r[1] : r[1], (TAU & 4) ? -r[2] : r[2] };
        float den = ref_ggx_denominator(us, n, -k.om_a2);
        d = core ? den : d;
    }
    float f;
    if (COUNT) {            // thr: this lane's acceptance threshold on the cosine (2 if its patch test failed)
        float qd = (t - k.cutoff) * __builtin_amdgcn_rcpf(d);
        f = (c * sq) * (qd * qd);
        bool ok = c > thr;
        f = ok ? f : 0.0f;
        n_acc += ok ? 1u : 0u;
    } else {
        // thr: this lane's cutoff (2 if its patch test failed).  The two acceptance tests of
        // cube.cpp:271 are folded into output clamps: t - cutoff <= 0 outside the window and
        // c * sqrt(t) <= 0 below the horizon both give weight 0, and the weight goes to 0
        // continuously at either edge, so no compare and no select are needed.
        float qd = sub_clamp01(t, thr) * __builtin_amdgcn_rcpf(d);
        f = mul_clamp01(c, sq) * (qd * qd);
    }
    const float fs = f * r[3];
#pragma unroll
    for (int so = 0; so < 8; so++) {
        const int si = so ^ TAU;
        acc[so][0] += fs;
        acc[so][1] = fmaf(f, r[4 + 3 * si], acc[so][1]);
        acc[so][2] = fmaf(f, r[5 + 3 * si], acc[so][2]);
        acc[so][3] = fmaf(f, r[6 + 3 * si], acc[so][3]);
    }
}

// Every wave works alone: 64 output octets, its own bounding cone, its own walk over
// the slice's patch octets.  Nothing is staged in LDS and there is no barrier: the
// records of a candidate patch octet are the same for all 64 lanes, so they are read
// with wave-uniform addresses (scalar loads through the scalar data cache) and feed
// the VALU as scalar operands.
#ifndef GO_MIN_WAVES
#define GO_MIN_WAVES 6   // 85 registers: the rarely touched FP64 sums spill to scratch, yet 6 waves per
                         // SIMD measured 7 % faster than 4 or 5 (more scalar-load latency hidden); 7 is slower
#endif
template <bool COUNT>
__global__ __launch_bounds__(GO_WG, GO_MIN_WAVES) void ggx_oct_pairs(
    const float4* __restrict__ rec, const float4* __restrict__ pn,
    uint32_t n_po, uint32_t ps2,
    uint32_t out_w, uint32_t tw, uint32_t first_block, uint32_t block_stride, uint32_t n_oct, uint32_t oct_local,
    uint32_t slices_and_order, uint32_t out_begin, uint32_t out_count,
    float cutoff, float a2m1, float cmin, float theta_acc, float core_thr,
    double* __restrict__ partial, unsigned long long* __restrict__ counts) {
    static_assert(GO_WG == 64, "one wave per workgroup: the grid below is counted in waves");
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    // Workgroups go to the 8 XCDs round-robin by their linear index, and every XCD has its own
    // L2.  Plain order is a (units, slices) grid, units fastest: the waves in flight then belong to one
    // slice if the call covers many units, to several if it covers few (one of 8 GPUs).  With
    // xcd_slices the grid is laid out so that XCD k only ever works on slices k, k + 8, ...:
    // the records an XCD's L2 has to hold are then those of one slice, however many are in
    // flight on the chip.  The host asks for it where that matters (narrow lobes, few units).
    // (One argument carries the slice count and, in its top bit, the order: scalar registers are
    // short in this kernel, and one more live argument sends a third of them through spills.)
    const uint32_t n_slices = slices_and_order & 0x7fffffffu;
    const bool xcd_slices = (slices_and_order >> 31) != 0u;
    const uint32_t units = (oct_local + GO_UNIT - 1) / GO_UNIT;
    uint32_t slice, wv;                                              // wv: this wave's unit among the call's
    if (xcd_slices) {                                                // n_slices is a multiple of 8 then
        const uint32_t seq = blockIdx.x / 8;
        slice = blockIdx.x % 8 + 8 * (seq / units);
        wv = seq % units;
    } else {                                                         // grid (units, slices)
        slice = blockIdx.y;
        wv = blockIdx.x;
    }
    const uint32_t lo = wv * GO_UNIT + lane;
    if (lo >= oct_local) return;                                     // whole wave: no barrier below
    const uint32_t gidx = go_global(lo, first_block, block_stride);
    const bool lane_valid = gidx < n_oct;
    const uint32_t oct_count = oct_local;
    const uint32_t hw = out_w / 2;
    uint32_t axis, oj, oi;
    go_decode(min(gidx, n_oct - 1), hw, tw, axis, oj, oi);

    // members of this octet inside the requested (face, y, x) range
    uint32_t member_mask = 0;
    if (lane_valid) {
#pragma unroll
        for (uint32_t m = 0; m < 8; m++) {
            Texel t = octet_member(axis, oj, oi, m, out_w, out_w);
            uint32_t o = (t.face * out_w + t.y) * out_w + t.x;
            if (o - out_begin < out_count) member_mask |= 1u << m;
        }
    }
    const bool active = member_mask != 0u;
    if (!__any(active)) return;      // no barrier anywhere below: a wave may leave alone

    Texel c0 = octet_member(axis, oj, oi, 0, out_w, out_w);
    const F3 n = ref_normalize(ref_normal_from(c0.face, c0.x, c0.y, out_w, out_w));   // cube.cpp:242
    GoConst kc;
    kc.cutoff = cutoff; kc.om_a2 = -a2m1; kc.cmin = cmin; kc.core_thr = core_thr;
    float a2p;
    {
        double eps = ((double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z) - 1.0;
        a2p = (float)((1.0 + (double)a2m1) - (double)kc.om_a2 * 0.75 * eps);
    }

    // Bounding cone of the wave's active canonical directions.
    float axx = active ? n.x : 0.f, axy = active ? n.y : 0.f, axz = active ? n.z : 0.f;
    for (int off = 32; off > 0; off >>= 1) {
        axx += __shfl_xor(axx, off); axy += __shfl_xor(axy, off); axz += __shfl_xor(axz, off);
    }
    {
        float inv = rsqrtf(axx * axx + axy * axy + axz * axz);
        axx *= inv; axy *= inv; axz *= inv;
    }
    float cmin_dot = active ? (axx * n.x + axy * n.y + axz * n.z) : 1.0f;
    for (int off = 32; off > 0; off >>= 1) cmin_dot = fminf(cmin_dot, __shfl_xor(cmin_dot, off));
    // A patch can pass some lane's test only within theta_acc of that lane's direction,
    // hence within theta_acc + cone radius of the axis (+ slack for rounding here).
    float wave_thr;
    {
        float ang = theta_acc + acosf(fminf(fmaxf(cmin_dot, -1.0f), 1.0f)) + 4e-3f;
        wave_thr = ang >= 3.14159f ? -2.0f : cosf(ang);
    }

    float acc[8][4];
    double dacc[8][4];
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
        for (int k = 0; k < 4; k++) { acc[m][k] = 0.0f; dacc[m][k] = 0.0; }
    uint32_t n_accepted = 0;
    unsigned long long tot_tested = 0, tot_accepted = 0;

    // Slices interleave the patch octets (slice s owns q = s, s + S, s + 2S, ...), so
    // every slice sees every region of the sphere and the workgroups of one output
    // tile carry equal shares of its cone.
    const uint32_t my_count = (n_po + n_slices - 1 - slice) / n_slices;

    for (uint32_t base = 0; base < my_count; base += 64) {
        // ---- candidates among the next 64 patch octets of the slice: sign patterns whose
        //      mirrored patch centre lies inside the wave's cone
        const uint32_t qi = base + lane;
        const uint32_t q = qi * n_slices + slice;
        uint32_t cmask = 0;
        if (qi < my_count) {
            const float4 c4 = pn[q];
            const float a = c4.x * axx, b = c4.y * axy, c = c4.z * axz;
#pragma unroll
            for (int tau = 0; tau < 8; tau++) {
                float dsum = ((tau & 1) ? -a : a) + ((tau & 2) ? -b : b) + ((tau & 4) ? -c : c);
                if (dsum >= wave_thr) cmask |= 1u << tau;
            }
        }
        unsigned long long bal = __ballot(cmask != 0u);
        while (bal) {                                  // ascending patch order
            const int src = __ffsll((long long)bal) - 1;
            bal &= bal - 1;
            const uint32_t po = __builtin_amdgcn_readlane(q, src);
            const uint32_t pmask = __builtin_amdgcn_readlane(cmask, src);
            // exact per-lane patch tests (cube.cpp:256-258), one per candidate pattern
            const float4 c4 = pn[po];
            float thr[8];
            uint32_t lane_pass = 0;
#pragma unroll
            for (int tau = 0; tau < 8; tau++) {
                bool pass = false;
                if (pmask & (1u << tau)) {    // wave-uniform
                    F3 pc = { (tau & 1) ? -c4.x : c4.x, (tau & 2) ? -c4.y : c4.y, (tau & 4) ? -c4.z : c4.z };
                    pass = active && ref_cone_test(pc, n, cutoff);
                }
                thr[tau] = pass ? (COUNT ? cmin : cutoff) : 2.0f;
                lane_pass |= pass ? (1u << tau) : 0u;
            }
            if (COUNT) tot_tested += (unsigned long long)__popc(lane_pass) * ps2 * 8ull;
            uint32_t wmask_v = 0;
#pragma unroll
            for (int tau = 0; tau < 8; tau++)
                if (__any((lane_pass >> tau) & 1u)) wmask_v |= 1u << tau;
            // the same in every lane; said so explicitly, the pattern tests in the record loop are
            // scalar bit tests instead of eight 64-bit lane masks that do not fit the scalar registers
            const uint32_t wmask = __builtin_amdgcn_readfirstlane(wmask_v);
            if (wmask == 0u) continue;

            const float4* __restrict__ rp = rec + (size_t)po * ps2 * GO_REC;   // wave-uniform
            // FP32 chunk: about 512 terms per accumulator (every record adds one term per
            // passing pattern).  wmask depends only on which octets share the wave, and
            // waves are fixed 64-octet groups of the level, so this choice is the same for
            // every way of splitting a bake.
            const uint32_t npat = __popc(wmask);
            const uint32_t flush = npat <= 2 ? 256u : (npat <= 4 ? 128u : (uint32_t)GO_FLUSH);
            for (uint32_t e0 = 0; e0 < ps2; e0 += flush) {
                const uint32_t e1 = min(ps2, e0 + flush);
                // Two scalar register sets: the next record is fetched while the current
                // one is evaluated (ping-pong, so nothing is copied).
                float4 ra[GO_REC], rb[GO_REC];
                auto load = [&](float4 (&dst)[GO_REC], uint32_t e) {
                    const float4* __restrict__ p = rp + (size_t)min(e, ps2 - 1) * GO_REC;
#pragma unroll
                    for (int qq = 0; qq < GO_REC; qq++) dst[qq] = p[qq];
                };
                auto unpack = [&](const float4 (&src)[GO_REC], float (&r)[28]) {
#pragma unroll
                    for (int qq = 0; qq < GO_REC; qq++) {
                        r[4 * qq] = src[qq].x; r[4 * qq + 1] = src[qq].y; r[4 * qq + 2] = src[qq].z; r[4 * qq + 3] = src[qq].w;
                    }
                };
                // Scalar loads return out of order, so the only wait is "all of them":
                // the first use of a record must come BEFORE the next record's loads are
                // issued, or that wait would cover the prefetch too.  touch() is that first
                // use; the scheduling barriers keep the order.
                auto touch = [&](const float4 (&src)[GO_REC]) {
                    float c = n.x * src[0].x;
                    c = fmaf(n.y, src[0].y, c);
                    c = fmaf(n.z, src[0].z, c);
                    return c;
                };
                auto eval_one = [&](const float4 (&src)[GO_REC], float c) {
                    float r[28];
                    unpack(src, r);
                    go_body<0, COUNT>(c, r, thr[0], a2p, kc, n, acc, n_accepted);
                };
                // the same for the walk over several patterns: the three products every pattern's
                // cosine is made of are the first use
                auto touch3 = [&](const float4 (&src)[GO_REC], float& a, float& b, float& cc) {
                    a = n.x * src[0].x; b = n.y * src[0].y; cc = n.z * src[0].z;
                    asm volatile("" :: "v"(a), "v"(b), "v"(cc));
                };
                auto eval_all = [&](const float4 (&src)[GO_REC], float a, float b, float cc) {
                    float r[28];
                    unpack(src, r);
                    const float s1 = a + b, s2 = a - b;
                    // The mask is re-read as an opaque scalar for every record: its eight bit tests
                    // then stay scalar compares inside the loop.  Hoisted out of it they become
                    // eight 64-bit lane masks, which do not fit the scalar registers, and every test
                    // pays two v_readlane and two more vector instructions, live pattern or not.
                    uint32_t wm = wmask;
                    asm volatile("" : "+s"(wm));
                    if (wm & 1u)   go_body<0, COUNT>(s1 + cc, r, thr[0], a2p, kc, n, acc, n_accepted);
                    if (wm & 2u)   go_body<1, COUNT>(cc - s2, r, thr[1], a2p, kc, n, acc, n_accepted);
                    if (wm & 4u)   go_body<2, COUNT>(s2 + cc, r, thr[2], a2p, kc, n, acc, n_accepted);
                    if (wm & 8u)   go_body<3, COUNT>(cc - s1, r, thr[3], a2p, kc, n, acc, n_accepted);
                    if (wm & 16u)  go_body<4, COUNT>(s1 - cc, r, thr[4], a2p, kc, n, acc, n_accepted);
                    if (wm & 32u)  go_body<5, COUNT>(-s2 - cc, r, thr[5], a2p, kc, n, acc, n_accepted);
                    if (wm & 64u)  go_body<6, COUNT>(s2 - cc, r, thr[6], a2p, kc, n, acc, n_accepted);
                    if (wm & 128u) go_body<7, COUNT>(-s1 - cc, r, thr[7], a2p, kc, n, acc, n_accepted);
                };
                load(ra, e0);
                if (wmask == 1u) {
                    for (uint32_t e = e0; e < e1; e += 2) {
                        const float ca = touch(ra);
                        __builtin_amdgcn_sched_barrier(0);
                        load(rb, e + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        eval_one(ra, ca);
                        const float cb = touch(rb);
                        __builtin_amdgcn_sched_barrier(0);
                        load(ra, e + 2);
                        __builtin_amdgcn_sched_barrier(0);
                        if (e + 1 < e1) eval_one(rb, cb);
                    }
                } else {
                    for (uint32_t e = e0; e < e1; e += 2) {
                        float a, b, cc;
                        touch3(ra, a, b, cc);
                        __builtin_amdgcn_sched_barrier(0);
                        load(rb, e + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        eval_all(ra, a, b, cc);
                        touch3(rb, a, b, cc);
                        __builtin_amdgcn_sched_barrier(0);
                        load(ra, e + 2);
                        __builtin_amdgcn_sched_barrier(0);
                        if (e + 1 < e1) eval_all(rb, a, b, cc);
                    }
                }
#pragma unroll
                for (int m = 0; m < 8; m++)
#pragma unroll
                    for (int k = 0; k < 4; k++) { dacc[m][k] += (double)acc[m][k]; acc[m][k] = 0.0f; }
            }
            if (COUNT) { tot_accepted += (unsigned long long)n_accepted * 8ull; n_accepted = 0; }
        }
    }

    // The 8 members of an octet see the same weights (each pairs with input member
    // so ^ tau), so their weight sums are one and the same number: stored once.
    if (lane_valid) {
#pragma unroll
        for (int m = 0; m < 8; m++)
#pragma unroll
            for (int k = 1; k < 4; k++)
                partial[((size_t)slice * GO_SUMS + (m * 3 + k - 1)) * oct_count + lo] = dacc[m][k];
        partial[((size_t)slice * GO_SUMS + 24) * oct_count + lo] = dacc[0][0];
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            tot_tested += __shfl_xor(tot_tested, off);
            tot_accepted += __shfl_xor(tot_accepted, off);
        }
        if (lane == 0) { atomicAdd(&counts[1], tot_tested); atomicAdd(&counts[2], tot_accepted); }
    }
}

// First stage of the slice sum when a level has many slices: group g adds its run of
// consecutive slices, in order, into one more set of partial sums.  One thread per
// (output octet, member, group).
__global__ __launch_bounds__(256) void ggx_oct_reduce(const double* __restrict__ partial, uint32_t n_slices,
                                                       uint32_t per_group, uint32_t n_oct, uint32_t first_block,
                                                       uint32_t block_stride, uint32_t oct_count,
                                                       double* __restrict__ reduced) {
    const uint32_t lo = blockIdx.x * GO_THREADS + threadIdx.x;
    if (lo >= oct_count) return;
    const uint32_t gidx = go_global(lo, first_block, block_stride);
    const uint32_t m = blockIdx.y, g = blockIdx.z;
    if (gidx >= n_oct) return;
    const uint32_t s0 = g * per_group, s1 = min(n_slices, s0 + per_group);
    // m < 8: the member's RGB sums; m == 8: the octet's weight sum
    const uint32_t j0 = m < 8 ? m * 3 : 24, nj = m < 8 ? 3 : 1;
    double v[3] = {0.0, 0.0, 0.0};
    for (uint32_t sl = s0; sl < s1; sl++) {
        for (uint32_t k = 0; k < nj; k++) v[k] += partial[((size_t)sl * GO_SUMS + (j0 + k)) * oct_count + lo];
    }
    for (uint32_t k = 0; k < nj; k++) reduced[((size_t)g * GO_SUMS + (j0 + k)) * oct_count + lo] = v[k];
}

// One thread per (output octet, member): coalesced over the octet index.
__global__ __launch_bounds__(256) void ggx_oct_final(const double* __restrict__ partial, uint32_t n_slices,
                                                      uint32_t out_w, uint32_t tw, uint32_t first_block,
                                                      uint32_t block_stride, uint32_t n_oct, uint32_t oct_count,
                                                      uint32_t out_begin, uint32_t out_count, float4* __restrict__ out) {
    const uint32_t lo = blockIdx.x * GO_THREADS + threadIdx.x;
    if (lo >= oct_count) return;
    const uint32_t gidx = go_global(lo, first_block, block_stride);
    const uint32_t m = blockIdx.y;
    if (gidx >= n_oct) return;
    uint32_t axis, oj, oi;
    go_decode(gidx, out_w / 2, tw, axis, oj, oi);
    Texel t = octet_member(axis, oj, oi, m, out_w, out_w);
    const uint32_t o = (t.face * out_w + t.y) * out_w + t.x;
    if (o - out_begin >= out_count) return;
    double v[3] = {0.0, 0.0, 0.0}, w = 0.0;
    for (uint32_t sl = 0; sl < n_slices; sl++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += partial[((size_t)sl * GO_SUMS + (m * 3 + k)) * oct_count + lo];
        w += partial[((size_t)sl * GO_SUMS + 24) * oct_count + lo];
    }
    double inv = 1.0 / w;   // cube.cpp:305 via linear.hpp:93-95
    out[o] = make_float4((float)(inv * v[0]), (float)(inv * v[1]), (float)(inv * v[2]), 0.0f);
}

// ---- launch plan -----------------------------------------------------------

bool ggx_octets_eligible(uint32_t in_w, uint32_t out_w, uint32_t patch) {
    if (patch == 0 || in_w % patch) return false;
    const uint32_t pw = in_w / patch;
    return cube_is_mirror_exact(in_w) && cube_is_mirror_exact(out_w) && cube_is_mirror_exact(pw);
}

GgxOctPlan ggx_oct_plan(uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips, uint32_t first_block,
                        uint32_t block_stride) {
    GgxOctPlan pl;
    pl.lv = ggx_level_params(mip, mips, in_w, in_w);
    const uint32_t ps = pl.lv.patch, pw = in_w / ps;
    pl.n_po = 3 * (pw / 2) * (pw / 2);
    pl.tw = go_tile_edge(out_w / 2);
    // slices come from the level geometry only (range/shard independence)
    const uint32_t n_oct_level = 3 * (out_w / 2) * (out_w / 2);
    const uint32_t units_level = (n_oct_level + GO_UNIT - 1) / GO_UNIT;      // one wave each per slice
    // Enough waves that an eighth of the level (one of 8 GPUs) still fills a GPU four
    // times over (4096 wave slots each): 131072, measured to be the best choice on one GPU
    // as well.  Large levels get up to eight times that (measured: 9 % faster on one GPU,
    // 0.89 -> 0.91 predicted efficiency on eight), as far as their work can carry the
    // partial sums (every (wave, slice) writes 64 x 25 doubles; that traffic is kept to a
    // few per cent of the estimated run time).
    const double cmin = fmax(0.0, 2.0 * (double)pl.lv.cutoff - 1.0);
    const double cone = 0.5 * (1.0 - cmin);                              // accepted share of the sphere
    const double wave_records = (double)n_oct_level / 64.0 * (double)pl.n_po * ps * ps * 8.0 * cone;
    const double seconds = wave_records * 70e-9 / 1024.0;                // ~70 ns per wave, record and pattern per SIMD
    double waves = 0.04 * seconds * 1.5e12 / (64.0 * GO_SUMS * 8.0);
    waves = fmin(fmax(waves, 131072.0), 1048576.0);
    uint32_t want = (uint32_t)ceil(waves / (double)units_level);
    // The waves in flight at one time belong to one slice and to output units from all over
    // the octant, on every XCD; together they read all of the slice's records, each many
    // times.  That set should sit in an XCD's 4 MiB L2 with room to spare (measured on the
    // 2048 -> 1024 level: 4 MiB per slice 1000 ms, 2 MiB 955 ms, 1 MiB 950 ms), as long as
    // the partial sums this costs stay below 32 GiB.
    {
        const double rec_total = (double)pl.n_po * ps * ps * GO_REC * sizeof(float4);
        const double per_slice = (double)units_level * GO_UNIT * GO_SUMS * sizeof(double);
        const double fit = ceil(rec_total / (double)(2u << 20));
        const double afford = floor((double)(32ull << 30) / per_slice);
        const double s = fmin(fit, afford);
        if (s > (double)want) want = (uint32_t)s;
    }
    if (const char* e = getenv("IBL_GGX_WAVES")) want = (uint32_t)ceil(atof(e) / (double)units_level);   // tuning knob
    uint32_t max_slices = pl.n_po;                           // at least one patch octet per slice
    if (max_slices == 0) max_slices = 1;
    uint32_t slices = want < 1 ? 1 : (want > max_slices ? max_slices : want);
    // a multiple of 8, so that the 8 XCDs can own equal shares of the slices (see ggx_oct_pairs)
    if (slices >= 16) slices = (slices + 7) / 8 * 8 <= max_slices ? (slices + 7) / 8 * 8 : slices / 8 * 8;
    pl.n_slices = slices;
    // units first_block, first_block + stride, ... below units_level
    pl.out_units = first_block < units_level ? (units_level - first_block + block_stride - 1) / block_stride : 0;
    const uint32_t oct_count = pl.out_units * GO_UNIT;
    // many slices are summed in two stages (fixed grouping, so still one summation order)
    pl.reduce_groups = pl.n_slices > 64 ? (pl.n_slices + 31) / 32 : 1;
    if (pl.reduce_groups > 64) pl.reduce_groups = 64;
    const uint32_t stage2 = pl.reduce_groups > 1 ? pl.reduce_groups : 0;
    pl.partial_bytes = (size_t)(pl.n_slices + stage2) * GO_SUMS * oct_count * sizeof(double);
    pl.rec_bytes = (size_t)pl.n_po * ps * ps * GO_REC * sizeof(float4);
    pl.pn_bytes = (size_t)pl.n_po * sizeof(float4);
    return pl;
}

hipError_t ggx_oct_launch(const GgxOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_w,
                          uint32_t first_block, uint32_t block_stride, uint32_t out_begin, uint32_t out_count,
                          float* d_out, void* d_rec, void* d_pn, double* d_partial, unsigned long long* d_counts,
                          hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t ps = pl.lv.patch, ps2 = ps * ps, pw = in_w / ps;
    const uint32_t n_rec = pl.n_po * ps2;
    ggx_oct_prep<<<(n_rec + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, ps, n_rec, (float4*)d_rec);
    ggx_oct_prep_patches<<<(pl.n_po + 255) / 256, 256, 0, stream>>>(pw, pl.n_po, (float4*)d_pn);
    const double cut = (double)pl.lv.cutoff;
    const float cmin = (float)fmax(0.0, 2.0 * cut - 1.0);
    const float theta_acc = (float)acos(fmin(1.0, fmax(-1.0, 2.0 * cut - 1.0)));
    const float a2m1 = pl.lv.alpha2 - 1.0f;
    const float core_thr = ggx_core_threshold(pl.lv.alpha2);
    // One wave per (unit, slice).  Slices are dealt to the XCDs (see ggx_oct_pairs) when several
    // slices would otherwise share each L2 -- the call's units do not fill the GPU's 6144 wave
    // slots -- and the level is in the narrow-lobe regime, where a record feeds one weight
    // evaluation and its load latency shows (2048 -> 1024 level, an eighth of the units: 150 -> 127 ms;
    // wide-lobe levels lose 1-3 % with it).
    const double cone = 0.5 * (1.0 - fmax(0.0, 2.0 * (double)pl.lv.cutoff - 1.0));
    uint32_t xcd_slices = (pl.n_slices % 8 == 0 && pl.out_units < 6144 && 8.0 * cone < 1.5) ? 1u : 0u;
    if (const char* e = getenv("IBL_GGX_XCD_SLICES")) xcd_slices = (atoi(e) != 0 && pl.n_slices % 8 == 0) ? 1u : 0u;
    const dim3 grid = xcd_slices ? dim3(pl.n_slices * pl.out_units) : dim3(pl.out_units, pl.n_slices);
    const uint32_t small_blocks = (pl.out_units * GO_UNIT + GO_THREADS - 1) / GO_THREADS;
    const uint32_t n_oct = 3 * (out_w / 2) * (out_w / 2);
    const uint32_t oct_count = pl.out_units * GO_UNIT;
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (d_counts)
        ggx_oct_pairs<true><<<grid, GO_WG, 0, stream>>>((const float4*)d_rec, (const float4*)d_pn, pl.n_po,
            ps2, out_w, pl.tw, first_block, block_stride, n_oct, oct_count, pl.n_slices | (xcd_slices << 31), out_begin, out_count, pl.lv.cutoff, a2m1, cmin,
            theta_acc, core_thr, d_partial, d_counts);
    else
        ggx_oct_pairs<false><<<grid, GO_WG, 0, stream>>>((const float4*)d_rec, (const float4*)d_pn, pl.n_po,
            ps2, out_w, pl.tw, first_block, block_stride, n_oct, oct_count, pl.n_slices | (xcd_slices << 31), out_begin, out_count, pl.lv.cutoff, a2m1, cmin,
            theta_acc, core_thr, d_partial, d_counts);
    if (ev1) (void)hipEventRecord(ev1, stream);
    const double* d_sums = d_partial;
    uint32_t n_sums = pl.n_slices;
    if (pl.reduce_groups > 1) {
        double* d_reduced = d_partial + (size_t)pl.n_slices * GO_SUMS * oct_count;
        const uint32_t per_group = (pl.n_slices + pl.reduce_groups - 1) / pl.reduce_groups;
        ggx_oct_reduce<<<dim3(small_blocks, 9, pl.reduce_groups), GO_THREADS, 0, stream>>>(
            d_partial, pl.n_slices, per_group, n_oct, first_block, block_stride, oct_count, d_reduced);
        d_sums = d_reduced;
        n_sums = pl.reduce_groups;
    }
    ggx_oct_final<<<dim3(small_blocks, 8), GO_THREADS, 0, stream>>>(d_sums, n_sums, out_w, pl.tw, first_block,
                                                            block_stride, n_oct, oct_count, out_begin, out_count,
                                                            (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
