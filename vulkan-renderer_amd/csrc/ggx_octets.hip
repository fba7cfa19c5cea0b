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
//   ggx_oct_pairs   one wave per workgroup and per (unit of 64 output octets in tile-major
//       order, run).  The patch octets of a level are dealt round-robin to the runs.  Per
//       64 patch octets a bounding-cone test marks, per lane, the candidate (patch octet,
//       sign patterns); the wave visits the candidates in patch order (ballot + readlane);
//       every lane runs the reference's exact float patch test for its own octet and each
//       candidate pattern; for every pattern that passes for any of its lanes the wave walks
//       the patch's records, reading them with wave-uniform (scalar) loads
//       (one walk per live pattern).  FP32 sums in registers cover up to 256 records; each is
//       then added to the run's FP32 running sums, which live in LDS.  No scratch memory, no
//       barriers.  Default for inputs of 64 x 64 and more (matrix form, ggx_mx_prep): the lanes
//       still evaluate the weights, 16 records at a time, and the 25 sums of those records are
//       three v_mfma_f32_32x32x16_bf16 per half of the wave on two bf16 pieces of every factor.
//   ggx_oct_reduce / ggx_oct_final   per output octet: the runs summed in a fixed two-stage
//       order in FP64, divided, scattered to the member texels that lie in the requested
//       (face, y, x) range.
// A texel's value does not depend on the range or shard a call covers: which texels a call
// stores decides only which waves run and what they store, never how a wave adds.
#include "ibl_device.h"
#include "ibl_kernels.h"
#include "ggx_common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace ibl {

constexpr int GO_UNIT = 64;             // output octets per wave: the unit of sharding
constexpr int GO_REC = 7;              // float4 per texel-octet record
constexpr int GO_SUMS = 25;             // doubles per (slice, output octet): 8 members x RGB + the weight sum

// Output octets are numbered tile-major: tiles of tw x tw canonical positions.
__host__ __device__ inline uint32_t go_tile_edge(uint32_t hw) {
    uint32_t tw = hw < 16 ? hw : 16;
    while (hw % tw) tw--;
    return tw;
}

__device__ __forceinline__ void go_decode(uint32_t g, uint32_t hw, uint32_t tw, uint32_t& axis, uint32_t& oj, uint32_t& oi) {
    octet_decode(g, hw, tw, axis, oj, oi);
}

// A call owns the 64-octet units units[0], units[1], ... of the tile-major octet order;
// lo indexes its own octets (and its partial sums).
__device__ __forceinline__ uint32_t go_global(uint32_t lo, const uint32_t* __restrict__ units) {
    return units[lo / GO_UNIT] * GO_UNIT + (lo % GO_UNIT);
}

// only_marked: the patch octets' table (ggx_oct_mark_patches); records of unmarked patch octets are
// then left unwritten (nobody reads them).
__global__ __launch_bounds__(256) void ggx_oct_prep(const float4* __restrict__ env, uint32_t w, uint32_t ps,
                                                     uint32_t n_rec, const float4* __restrict__ only_marked,
                                                     float4* __restrict__ rec) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rec) return;
    const uint32_t ps2 = ps * ps, hpw = (w / ps) / 2;
    const uint32_t po = k / ps2, sub = k - po * ps2;
    if (only_marked && only_marked[po].w == 0.0f) return;
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

// Behind the n_po centres: one float4 whose .x holds the number of marked patch octets (as bits), then
// their list (ggx_oct_mark_patches, ggx_oct_list_marks).
__global__ __launch_bounds__(256) void ggx_oct_prep_patches(uint32_t pw, uint32_t n_po, float4* __restrict__ pn) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) pn[n_po] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p >= n_po) return;
    const uint32_t hpw = pw / 2;
    const uint32_t axis = p / (hpw * hpw);
    const uint32_t rem = p - axis * (hpw * hpw);
    const uint32_t pj = rem / hpw, pi = rem - pj * hpw;
    Texel c = octet_member(axis, pj, pi, 0, pw, pw);
    F3 u = ref_normalize(ref_normal_from(c.face, c.x, c.y, pw, pw));   // cube.cpp:198
    pn[p] = make_float4(u.x, u.y, u.z, 0.0f);
}

// Patches whose radiance sits in a few texels (a sun of a few texels, one lamp on black) are marked
// (pn[p].w = 1) and counted (pn[n_po].x).  The matrix form multiplies 16-bit factors; the 2^-18 cut of
// a term is random and averages out over the texels under a lobe, except where one texel IS the sum.
// For a patch P and one channel, with S1 = sum of L and S2 = sum of L^2 over its texels, the terms of P
// contribute at most eps^2 W^2 S2 to an output's squared error and W S1 to its value (W: the lobe's
// weight over the patch, nearly constant at the scale of a patch), so if S2 <= S1^2 / n_min holds for
// every patch that takes the matrix form, every output is within eps / sqrt(n_min) whatever the lobe.
// Patch octets with a member patch that fails this in some channel are walked on the vector pipe with
// FP32 products instead (ggx_oct_pairs<.., MARKS = 2>), so a lone light gets the vector form's bits.
// One workgroup per patch octet: 32 threads per member, FP64 sums (L^2 of an RGBE value can leave FP32).
__global__ __launch_bounds__(256) void ggx_oct_mark_patches(const float4* __restrict__ env, uint32_t w, uint32_t ps,
                                                             uint32_t n_po, float n_min, float4* __restrict__ pn) {
    const uint32_t po = blockIdx.x;
    const uint32_t m = threadIdx.x >> 5, l = threadIdx.x & 31u;
    const uint32_t ps2 = ps * ps, hpw = (w / ps) / 2;
    const uint32_t axis = po / (hpw * hpw);
    const uint32_t prem = po - axis * (hpw * hpw);
    const uint32_t pj = prem / hpw, pi = prem - pj * hpw;
    double s1[3] = {0.0, 0.0, 0.0}, s2[3] = {0.0, 0.0, 0.0};
    for (uint32_t sub = l; sub < ps2; sub += 32) {
        const uint32_t sy = sub / ps, sx = sub - sy * ps;
        Texel t = octet_member(axis, pj * ps + sy, pi * ps + sx, m, w, w);
        const float4 L = env[(t.face * w + t.y) * w + t.x];
        const double v[3] = { (double)L.x, (double)L.y, (double)L.z };
#pragma unroll
        for (int k = 0; k < 3; k++) { s1[k] += v[k]; s2[k] += v[k] * v[k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int off = 16; off > 0; off >>= 1) { s1[k] += __shfl_xor(s1[k], off); s2[k] += __shfl_xor(s2[k], off); }
    bool few = false;
#pragma unroll
    for (int k = 0; k < 3; k++) few = few || (s2[k] * (double)n_min > s1[k] * s1[k]);
    if (__syncthreads_or(few ? 1 : 0) && threadIdx.x == 0) ((float*)(pn + po))[3] = 1.0f;
}

// The marked patch octets in ascending order behind the centres, and their number: one workgroup, a
// running count over blocks of 1024 patch octets (the order must not depend on who got there first:
// it decides which run a patch's terms are added in).
__global__ __launch_bounds__(1024) void ggx_oct_list_marks(uint32_t n_po, float4* __restrict__ pn) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    uint32_t* __restrict__ list = (uint32_t*)(pn + n_po + 1);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t b = 0; b < n_po; b += 1024) {
        const uint32_t p = b + threadIdx.x;
        const bool m = p < n_po && pn[p].w != 0.0f;
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
    if (threadIdx.x == 0) ((uint32_t*)(pn + n_po))[0] = s_base;
}

// A record in scalar registers, and its loads and waits written out.  The outputs are
// early-clobber (they are written long after the instruction has issued, so they must not share
// a register with the address); go_rec_wait ties the registers to the wait, so that no use can
// be scheduled in front of it.
struct GoRec { go_u32x4 a; go_u32x16 b; go_u32x8 c; };     // float4 0 | 1-4 | 5-6 of the record

__device__ __forceinline__ void go_rec_load(GoRec& r, const float4* __restrict__ p) {
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x10\n\ts_load_dwordx8 %2, %3, 0x50"
                 : "=&s"(r.a), "=&s"(r.b), "=&s"(r.c) : "s"(p));
}
__device__ __forceinline__ void go_rec_wait(GoRec& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b), "+s"(r.c));
}
__device__ __forceinline__ void go_unpack(const GoRec& s, float (&r)[28]) {
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = __uint_as_float(s.a[i]);
#pragma unroll
    for (int i = 0; i < 16; i++) r[4 + i] = __uint_as_float(s.b[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) r[20 + i] = __uint_as_float(s.c[i]);
}

// ---- matrix form of the records (ggx_oct_pairs<.., MX = true>) ----------------------------------
// The sums of a walk are a small dense product: (64 output octets x records) weights times (records
// x 25) record columns -- the 24 member radiances and |w|^-3.  The matrix form feeds it to
// v_mfma_f32_32x32x16_bf16 with every FP32 factor cut into two bf16 pieces (nearest-even upper
// piece, then the nearest-even bf16 of the exact remainder: 16 significant bits, relative error
// <= 2^-18) and the three partial products hi*hi, hi*mid, mid*hi added in the FP32 accumulators.
// A weight's two pieces multiply all 25 columns of its record, so what the cut does to a weight
// cancels in the quotient sum(L W) / sum(W) (cube.cpp:305) to first order; what it does to a record
// column is a fixed 2^-18 perturbation of that input texel.
// Layout: canonical unit direction and |w|^-3 of the records, patch-octet-major like the vector
// form, four records to a 64-byte block (x0-3, y0-3, z0-3, |w|^-3 0-3: neighbouring records' components
// are then scalar register pairs for packed instructions); then per 16 consecutive records two 1 KB tiles (upper pieces, lower pieces) in the B-operand
// order of the instruction: lane l holds column l % 32 of records 8 (l / 32) + 0..7 as 8 bf16.
constexpr int GO_MX_REC_BYTES = 16 + 2 * 64;          // direction + two pieces of 32 columns

__global__ __launch_bounds__(256) void ggx_mx_prep(const float4* __restrict__ env, uint32_t w, uint32_t ps,
                                                    uint32_t n_rec, float4* __restrict__ dirs, go_u32x4* __restrict__ tiles) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = t / 64, l = t % 64;
    if (grp * GO_MX_GROUP >= n_rec) return;
    const uint32_t col = l & 31u, half = l >> 5;
    const uint32_t ps2 = ps * ps, hpw = (w / ps) / 2;
    uint32_t hi[8], mid[8];
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) {
        const uint32_t k = grp * GO_MX_GROUP + half * 8 + e;
        const uint32_t po = k / ps2, sub = k - po * ps2;
        const uint32_t sy = sub / ps, sx = sub - sy * ps;
        const uint32_t axis = po / (hpw * hpw);
        const uint32_t prem = po - axis * (hpw * hpw);
        const uint32_t pj = prem / hpw, pi = prem - pj * hpw;
        const uint32_t j = pj * ps + sy, i = pi * ps + sx;
        Texel c = octet_member(axis, j, i, 0, w, w);
        F3 d = ref_normal_from(c.face, c.x, c.y, w, w);     // all components > 0
        float l2 = ref_dot(d, d);
        float s = 1.0f / (l2 * sqrtf(l2));
        float v = 0.0f;
        if (col < 24u) {
            const uint32_t m = col / 3u, ch = col - 3u * m;
            Texel tx = octet_member(axis, j, i, m, w, w);
            const float4 L = env[(tx.face * w + tx.y) * w + tx.x];
            v = (ch == 0u ? L.x : ch == 1u ? L.y : L.z) * s;
        } else if (col == 24u) {
            v = s;
        }
        hi[e] = go_bf16_bits(v);
        mid[e] = go_bf16_bits(v - __uint_as_float(hi[e] << 16));
        if (col == 0u) {      // four records to a 64-byte block: their x, their y, their z, their |w|^-3
            F3 u = ref_normalize(d);
            float* blk = (float*)(dirs + (k & ~3u));
            blk[k & 3u] = u.x; blk[4 + (k & 3u)] = u.y; blk[8 + (k & 3u)] = u.z; blk[12 + (k & 3u)] = s;
        }
    }
    go_u32x4 a, b;
#pragma unroll
    for (int q = 0; q < 4; q++) { a[q] = hi[2 * q] | (hi[2 * q + 1] << 16); b[q] = mid[2 * q] | (mid[2 * q + 1] << 16); }
    tiles[(size_t)(grp * 2) * 64 + l] = a;
    tiles[(size_t)(grp * 2 + 1) * 64 + l] = b;
}


template <bool COUNT, bool CORE>
__device__ __forceinline__ void go_body(float c, const float (&r)[28], float thr, float a2p, const GoConst& k,
                                        const F3& nt, float (&acc)[8][3], float& accw, uint32_t& n_acc) {
    const float f = go_weight<COUNT, CORE>(c, r[0], r[1], r[2], thr, a2p, k, nt, n_acc);
    accw = fmaf(f, r[3], accw);
#pragma unroll
    for (int si = 0; si < 8; si++) {
        acc[si][0] = fmaf(f, r[4 + 3 * si], acc[si][0]);
        acc[si][1] = fmaf(f, r[5 + 3 * si], acc[si][1]);
        acc[si][2] = fmaf(f, r[6 + 3 * si], acc[si][2]);
    }
}

// Every wave works alone: 64 output octets, its own bounding cone, its own walk over
// the run's patch octets.  Records are not staged in LDS and there is no barrier: the
// records of a candidate patch octet are the same for all 64 lanes, so they are read
// with wave-uniform addresses (scalar loads through the scalar data cache) and feed
// the VALU as scalar operands.
#ifndef GO_MIN_WAVES
#define GO_MIN_WAVES 6
#endif
// EARLY (wide lobes: the acceptance window reaches the horizon): a record that no lane of the
// wave accepts is dropped after its cosine.  The reference's patch test keeps whole 16 x 16 patches
// that straddle the window's edge, and at these levels a third of the texels it then tests fail
// (cube.cpp:271); a wave's 64 directions lie within a few degrees, so they mostly fail together.
// Narrow lobes accept 99 % of what the patch test lets through and skip the test.
// MX: the sums of a walk go through the matrix pipe (see ggx_mx_prep for the operand format and
// what it costs in accuracy); rec then points at the records' directions, followed by their tiles.
#ifndef GO_MX_MIN_WAVES
#define GO_MX_MIN_WAVES 4          // 128 registers: no spills (5: 96 and 308 bytes of scratch, 2.5 % slower)
#endif
#ifndef GO_MX_TERMS
#define GO_MX_TERMS 3                    // 3: hi*hi + hi*mid + mid*hi;  4: + mid*mid
#endif
#ifndef GO_CUT_DOT2
#define GO_CUT_DOT2 1                    // lower pieces of the weights through v_dot2c_f32_bf16 (0: shift, mask, FMA)
#endif
// MARKS (see ggx_oct_mark_patches): 0 every patch octet; 1 all but the marked ones; 2 only the marked ones,
// and the run's sums are ADDED to what the launch before (MARKS = 1, same grid) stored.
template <bool COUNT, bool EARLY, bool MX = false, bool NARROW = false, int MARKS = 0>
__global__ __launch_bounds__(GO_WG, COUNT ? 4 : (MX ? GO_MX_MIN_WAVES : GO_MIN_WAVES)) void ggx_oct_pairs(
    const float4* __restrict__ rec, const float4* __restrict__ pn, const uint32_t* __restrict__ units,
    uint32_t n_po, uint32_t ps2,
    uint32_t out_w, uint32_t tw, uint32_t n_oct, uint32_t oct_local,
    uint32_t runs_and_order, uint32_t out_begin, uint32_t out_count,
    float cutoff, float a2m1, float cmin, float theta_acc, float core_thr,
    float* __restrict__ partial, unsigned long long* __restrict__ counts) {
    static_assert(GO_WG == 64, "one wave per workgroup: the grid below is counted in waves");
    static_assert(!(MX && (COUNT || EARLY)), "the matrix form neither counts nor drops records");
    static_assert(MX || !NARROW, "NARROW selects the weight form of the matrix kernel");
    static_assert(MARKS == 0 || (!COUNT && (MX == (MARKS == 1))), "marked patch octets: matrix form without them, vector form on them");
    // the run's running sums: [sum][lane] (matrix form: rows padded, its fold walks down the sums)
    __shared__ float s_run[GO_SUMS][MX ? GO_WG + 1 : GO_WG];
    const uint32_t lane = threadIdx.x & 63;
    // Workgroups go to the 8 XCDs round-robin by their linear index, and every XCD has its own
    // L2.  Plain order is (run, unit) with units fastest: the waves in flight then belong to one
    // run if the call covers many units, to several if it covers few (one of 8 GPUs).  With
    // xcd_runs the grid is laid out so that XCD k only ever works on runs k, k + 8, ...:
    // the records an XCD's L2 has to hold are then those of one run's current subset, however
    // many runs are in flight on the chip.  The host asks for it where that matters (narrow
    // lobes, few units).
    // (One argument carries the level's run count, the first run of this launch and the order:
    // scalar registers are short in this kernel.)
    const uint32_t n_runs = runs_and_order & 0x7fffu;                // runs of the level
    const uint32_t run_base = (runs_and_order >> 15) & 0x7fffu;      // first run of this launch
    const bool xcd_runs = (runs_and_order >> 31) != 0u;
    const uint32_t n_units = oct_local / GO_UNIT;
    uint32_t run_local, wv;                                          // wv: this wave's unit among the call's
    if (xcd_runs) {                                                  // the launch's run count is a multiple of 8 then
        const uint32_t seq = blockIdx.x / 8;
        run_local = blockIdx.x % 8 + 8 * (seq / n_units);
        wv = seq % n_units;
    } else {
        run_local = blockIdx.x / n_units;
        wv = blockIdx.x % n_units;
    }
    const uint32_t run = run_base + run_local;
    // fewer items (marked patch octet, pattern, stretch of records) than runs: the sums of this run stand as they are
    if (MARKS == 2 && run >= 8u * ((ps2 + GO_MARK_PART - 1) / GO_MARK_PART) * __float_as_uint(pn[n_po].x)) return;
    const uint32_t lo = wv * GO_UNIT + lane;
    const uint32_t gidx = go_global(lo, units);
    const bool lane_valid = gidx < n_oct;
    const uint32_t hw = out_w / 2;
    uint32_t axis, oj, oi;
    go_decode(min(gidx, n_oct - 1), hw, tw, axis, oj, oi);

    // Members of this octet inside the requested (face, y, x) range.  They decide whether the
    // wave has anything to store, and nothing else: a wave that runs adds for all its octets,
    // so that its sums do not depend on the range.
    {
        bool wanted = false;
        if (lane_valid) {
#pragma unroll
            for (uint32_t m = 0; m < 8; m++) {
                Texel t = octet_member(axis, oj, oi, m, out_w, out_w);
                uint32_t o = (t.face * out_w + t.y) * out_w + t.x;
                wanted = wanted || (o - out_begin < out_count);
            }
        }
        if (!__any(wanted)) return;      // no barrier anywhere below: a wave may leave alone
    }

    Texel c0 = octet_member(axis, oj, oi, 0, out_w, out_w);
    const F3 n = ref_normalize(ref_normal_from(c0.face, c0.x, c0.y, out_w, out_w));   // cube.cpp:242
    GoConst kc;
    kc.cutoff = cutoff; kc.om_a2 = -a2m1; kc.cmin = cmin; kc.core_thr = core_thr;
    float a2p;
    {
        double eps = ((double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z) - 1.0;
        a2p = (float)((1.0 + (double)a2m1) - (double)kc.om_a2 * 0.75 * eps);
    }

    // Bounding cone of the wave's canonical directions.
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
    // A patch can pass some lane's test only within theta_acc of that lane's direction,
    // hence within theta_acc + cone radius of the axis (+ slack for rounding here).
    // A record can lie in some lane's lobe core (d < core_thr) only if its patch centre is
    // within the core's angle + the patch's angular radius + the cone radius of the axis.
    float wave_thr, core_cos;
    {
        const float radius = acosf(fminf(fmaxf(cmin_dot, -1.0f), 1.0f));
        float ang = theta_acc + radius + 4e-3f;
        wave_thr = ang >= 3.14159f ? -2.0f : cosf(ang);
        core_cos = 2.0f;                                         // no core: no patch qualifies
        if (core_thr > 0.0f) {
            // d = u (1 - a2) + a2 (+ a term of order 1e-7) with u = (1 - cos)/2
            const float u_core = (core_thr - (1.0f + a2m1)) / kc.om_a2 + 1e-5f;
            const float th_core = acosf(fminf(fmaxf(1.0f - 2.0f * u_core, -1.0f), 1.0f));
            const float pw = 2.0f * sqrtf((float)n_po * (1.0f / 3.0f));      // patches per face edge
            ang = th_core + 1.4143f / pw + radius + 4e-3f;
            core_cos = ang >= 3.14159f ? -2.0f : cosf(ang);
        }
    }
    // The cone is the same in every lane: kept in scalar registers (as vector registers these
    // rarely used values are the first the register allocator sends to scratch memory).
    auto uniform = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    axx = uniform(axx); axy = uniform(axy); axz = uniform(axz);
    wave_thr = uniform(wave_thr); core_cos = uniform(core_cos);

    // bf16 pairs (-1, -0) and (-0, -1).  (With +0 the first is the bit pattern of an inline constant,
    // which this instruction does not read as a bf16 pair: results were off by 1e-2.)
    constexpr uint32_t cut_lo = 0x8000bf80u, cut_hi = 0xbf808000u;
    float acc[8][3], accw = 0.0f;     // FP32 chunk sums, indexed by input member
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
        for (int k = 0; k < 3; k++) acc[m][k] = 0.0f;
    // Matrix form: the chunk sums are two 32 x 32 accumulator tiles (output octets 0-31 and 32-63 of
    // the wave x record columns); this lane holds column lane % 32 of both, rows (i % 4) + 8 (i / 4) +
    // 4 (lane / 32) in element i.
    go_f32x16 mx0, mx1;
#pragma unroll
    for (int i = 0; i < 16; i++) { mx0[i] = 0.0f; mx1[i] = 0.0f; }
    const uint32_t mx_col = lane & 31u, mx_member = mx_col / 3u, mx_chan = mx_col - 3u * mx_member;
#pragma unroll
    for (int k = 0; k < GO_SUMS; k++) s_run[k][lane] = 0.0f;
    uint32_t n_accepted = 0;
    unsigned long long tot_tested = 0, tot_accepted = 0, tot_walked = 0, tot_evaluated = 0, tot_visited = 0;

    // Three levels of sums, all in a fixed order: FP32 chunks of at most 256 terms in registers;
    // the run's FP32 running sums of its chunks (a few dozen) in LDS, indexed by output member;
    // the runs (dozens to hundreds per output octet) in FP64 by ggx_oct_reduce / ggx_oct_final.
    // What the FP32 running sums cost is the rounding of a few dozen additions per run, about
    // 1e-7 of a run's sum and, runs being independent, a few 1e-8 of the total: the size of the
    // per-term rounding the FP32 weights carry anyway.  The chunk was gathered under sign pattern
    // tau: input member si belongs to output member si ^ tau.  The eight members of an octet see
    // the same weights, so the weight sum is kept once.
    auto fold = [&](auto tau_c) __attribute__((always_inline)) {
        constexpr int TAU = decltype(tau_c)::value;
#pragma unroll
        for (int si = 0; si < 8; si++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s_run[(si ^ TAU) * 3 + k][lane] += acc[si][k];
                acc[si][k] = 0.0f;
            }
        s_run[24][lane] += accw;
        accw = 0.0f;
    };
    auto flush = [&](uint32_t tau) __attribute__((always_inline)) {          // wave-uniform
        if constexpr (MX) {
            // column mx_col of the tiles is sum mx_chan of input member mx_member (24: the weight sum)
            if (mx_col < 25u) {
                const uint32_t oc = mx_col < 24u ? (mx_member ^ tau) * 3u + mx_chan : 24u;
                float* p = &s_run[oc][4u * (lane >> 5)];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int row = (i & 3) + 8 * (i >> 2);
                    p[row] += mx0[i];
                    p[32 + row] += mx1[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) { mx0[i] = 0.0f; mx1[i] = 0.0f; }
            return;
        }
        switch (tau) {
            case 0: fold(std::integral_constant<int, 0>{}); break;
            case 1: fold(std::integral_constant<int, 1>{}); break;
            case 2: fold(std::integral_constant<int, 2>{}); break;
            case 3: fold(std::integral_constant<int, 3>{}); break;
            case 4: fold(std::integral_constant<int, 4>{}); break;
            case 5: fold(std::integral_constant<int, 5>{}); break;
            case 6: fold(std::integral_constant<int, 6>{}); break;
            default: fold(std::integral_constant<int, 7>{}); break;
        }
    };
    // An FP32 chunk holds at most 256 terms per sum, all gathered under one pattern:
    // consecutive walks under the same pattern share a chunk while they fit.
    uint32_t acc_tau = 0, acc_terms = 0;      // wave-uniform
    uint32_t walked_any = 0;                  // wave-uniform (MARKS == 2)

    // One candidate patch octet: pmask holds the sign patterns whose mirrored patch centre lies inside the
    // wave's cone (bits 0-7) and, of those, the ones close enough for the lobe core (bits 8-15).
    // The records of the patch octet a walk covers: all of them, or (MARKS == 2) one stretch.
    uint32_t w_begin = 0, w_count = ps2;      // wave-uniform
    auto visit = [&](uint32_t po, uint32_t pmask) __attribute__((always_inline)) {
        // exact per-lane patch tests (cube.cpp:256-258), one per candidate pattern
        const float4 c4 = pn[po];
        uint32_t lane_pass = 0;
#pragma unroll
        for (int tau = 0; tau < 8; tau++) {
            if (pmask & (1u << tau)) {    // wave-uniform
                F3 pc = { (tau & 1) ? -c4.x : c4.x, (tau & 2) ? -c4.y : c4.y, (tau & 4) ? -c4.z : c4.z };
                if (lane_valid && ref_cone_test(pc, n, cutoff)) lane_pass |= 1u << tau;
            }
        }
        if (COUNT) tot_tested += (unsigned long long)__popc(lane_pass) * ps2 * 8ull;
        uint32_t wmask_v = 0;
#pragma unroll
        for (int tau = 0; tau < 8; tau++)
            if (__any((lane_pass >> tau) & 1u)) wmask_v |= 1u << tau;
        uint32_t wmask = __builtin_amdgcn_readfirstlane(wmask_v);    // the same in every lane

        if (COUNT && wmask != 0u) tot_visited += ps2;
        if (MARKS == 2) walked_any |= wmask;
        const float4* __restrict__ rp = rec + ((size_t)po * ps2 + w_begin) * GO_REC;   // wave-uniform
        // ---- one walk over the patch octet's records per live pattern
        while (wmask) {
            const uint32_t tau = (uint32_t)__builtin_ctz(wmask);
            wmask &= wmask - 1;
            if (acc_terms != 0u && (acc_tau != tau || acc_terms + w_count > (uint32_t)GO_CHUNK)) { flush(acc_tau); acc_terms = 0u; }
            acc_tau = tau;
            acc_terms += w_count;
            // this lane's cutoff for the walk (2 if its patch test failed: weight 0)
            const float thr = ((lane_pass >> tau) & 1u) ? (COUNT ? cmin : cutoff) : 2.0f;
            // (EARLY) the lanes whose patch test passed, as a scalar mask: a register per lane
            // for it would be the one too many
            const unsigned long long pass_mask = EARLY ? __ballot(((lane_pass >> tau) & 1u) != 0u) : 0ull;
            const F3 nt = { __uint_as_float(__float_as_uint(n.x) ^ ((tau & 1u) << 31)),
                            __uint_as_float(__float_as_uint(n.y) ^ ((tau & 2u) << 30)),
                            __uint_as_float(__float_as_uint(n.z) ^ ((tau & 4u) << 29)) };
            const bool core_near = ((pmask >> (8u + tau)) & 1u) != 0u;       // wave-uniform
            // Two scalar register sets (ping-pong, so nothing is copied): the next record is
            // fetched while the current one is evaluated.  The loads and their waits are
            // written out (GoRec): scalar loads return out of order, so the only wait is
            // "all of them", and left to the compiler both the loads and the wait end up
            // right in front of the first use.
            GoRec ra, rb;
            auto walk = [&](auto core_c) __attribute__((always_inline)) {
                constexpr bool CORE = decltype(core_c)::value;
                auto eval = [&](const GoRec& src) __attribute__((always_inline)) {
                    float r[28];
                    go_unpack(src, r);
                    float c = nt.x * r[0];
                    c = fmaf(nt.y, r[1], c);
                    c = fmaf(nt.z, r[2], c);
                    if (EARLY && (__ballot(c > cmin) & pass_mask) == 0ull) return;      // weight 0 in every lane
                    if (COUNT) tot_evaluated += 1;
                    go_body<COUNT, CORE>(c, r, thr, a2p, kc, nt, acc, accw, n_accepted);
                };
                go_rec_load(ra, rp);
                for (uint32_t e = 0; e < w_count; e += 2) {
                    go_rec_wait(ra);
                    go_rec_load(rb, rp + (size_t)min(e + 1, w_count - 1) * GO_REC);
                    eval(ra);
                    go_rec_wait(rb);
                    go_rec_load(ra, rp + (size_t)min(e + 2, w_count - 1) * GO_REC);
                    if (e + 1 < w_count) eval(rb);
                }
                go_rec_wait(ra);       // nothing of this walk is in flight when it ends
            };
            auto walk_mx = [&](auto core_c) __attribute__((always_inline)) {
                constexpr bool CORE = decltype(core_c)::value;
                const float4* __restrict__ dp = rec + (size_t)po * ps2;                 // wave-uniform
                const go_u32x4* __restrict__ tp = (const go_u32x4*)(rec + (size_t)n_po * ps2)
                                                  + (size_t)po * (ps2 / GO_MX_GROUP) * 128 + lane;
                GoDir4 da, db;
                // four records: weights, cut into their two bf16 pieces, two records to a register
                auto quad = [&](const GoDir4& src, uint32_t& h0, uint32_t& h1, uint32_t& m0, uint32_t& m1)
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
                        // near the lobe core: record by record (two exact denominators side by side cost
                        // more registers than 4 waves per SIMD leave)
                        uint32_t none = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float ux = __uint_as_float(src.v[e]), uy = __uint_as_float(src.v[4 + e]),
                                        uz = __uint_as_float(src.v[8 + e]);
                            float c = nt.x * ux;
                            c = fmaf(nt.y, uy, c);
                            c = fmaf(nt.z, uz, c);
                            f[e] = go_weight<false, true>(c, ux, uy, uz, thr, a2p, kc, nt, none);
                            // (left alone the compiler evaluates cosine, u and d of all 16 records and their
                            // exact denominators first and keeps them for the weights: 40 registers more than
                            // the rest of the kernel needs; the empty statement ends a record before the next)
                            asm volatile("" : "+v"(f[e]));
                        }
                    }
                    h0 = go_pack_bf16(f[0], f[1]);
                    h1 = go_pack_bf16(f[2], f[3]);
                    if constexpr (GO_CUT_DOT2 && !CORE) {
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
                go_dir_load(da, dp);
                for (uint32_t g = 0; g < ps2; g += GO_MX_GROUP) {
                    const go_u32x4 b_hi = tp[0], b_mid = tp[64];
                    tp += 128;
                    uint32_t ah[8], am[8];
                    go_dir_wait(da);
                    go_dir_load(db, dp + g + 4);
                    quad(da, ah[0], ah[1], am[0], am[1]);
                    go_dir_wait(db);
                    go_dir_load(da, dp + g + 8);
                    quad(db, ah[2], ah[3], am[2], am[3]);
                    go_dir_wait(da);
                    go_dir_load(db, dp + g + 12);
                    quad(da, ah[4], ah[5], am[4], am[5]);
                    go_dir_wait(db);
                    go_dir_load(da, dp + min(g + 16, ps2 - 4));
                    quad(db, ah[6], ah[7], am[6], am[7]);
                    // A operands: rows are the output octets, so the upper lanes' records 0-7 and the lower
                    // lanes' records 8-15 change places (lanes 32-63 of the first register with lanes 0-31
                    // of the second)
                    go_u32x4 a0h, a1h, a0m, a1m;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        auto sh = __builtin_amdgcn_permlane32_swap(ah[i], ah[4 + i], false, false);
                        a0h[i] = sh[0]; a1h[i] = sh[1];
                        auto sm = __builtin_amdgcn_permlane32_swap(am[i], am[4 + i], false, false);
                        a0m[i] = sm[0]; a1m[i] = sm[1];
                    }
                    auto bf = [](const go_u32x4& v) { return __builtin_bit_cast(go_bf16x8, v); };
                    mx0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0m), bf(b_hi), mx0, 0, 0, 0);
                    mx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1m), bf(b_hi), mx1, 0, 0, 0);
                    if (GO_MX_TERMS >= 4) {
                        mx0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0m), bf(b_mid), mx0, 0, 0, 0);
                        mx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1m), bf(b_mid), mx1, 0, 0, 0);
                    }
                    mx0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0h), bf(b_mid), mx0, 0, 0, 0);
                    mx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1h), bf(b_mid), mx1, 0, 0, 0);
                    mx0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0h), bf(b_hi), mx0, 0, 0, 0);
                    mx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1h), bf(b_hi), mx1, 0, 0, 0);
                }
                go_dir_wait(da);       // nothing of this walk is in flight when it ends
            };
            if constexpr (MX) {
                if (core_near) walk_mx(std::true_type{});
                else walk_mx(std::false_type{});
            } else {
                if (core_near) walk(std::true_type{});
                else walk(std::false_type{});
            }
            if (COUNT) { tot_accepted += (unsigned long long)n_accepted * 8ull; n_accepted = 0; tot_walked += ps2; }
        }
    };

    if constexpr (MARKS == 2) {
        // The marked patch octets, in ascending order, times the 8 sign patterns times stretches of 64
        // records are dealt to the runs one item at a time: a handful of marked patches -- the usual case --
        // is then one short walk for each of many waves, not eight long ones after each other for a few (a
        // lone wave waits out the latency of every record it loads).  Items are numbered pattern-major: with
        // the pattern in the low bits, the patches a unit's cone holds under one pattern would all fall to
        // runs of one residue modulo 8, which the grid order places on one XCD.
        const uint32_t* __restrict__ list = (const uint32_t*)(pn + n_po + 1);
        const uint32_t parts = (ps2 + GO_MARK_PART - 1) / GO_MARK_PART;
        const uint32_t per_pattern = parts * __float_as_uint(pn[n_po].x), n_items = 8u * per_pattern;
        for (uint32_t it = run; it < n_items; it += n_runs) {
            const uint32_t tau = it / per_pattern, within = it - tau * per_pattern;
            const uint32_t part = within % parts, po = list[within / parts];
            w_begin = part * GO_MARK_PART;
            w_count = min(ps2 - w_begin, (uint32_t)GO_MARK_PART);
            const float4 c4 = pn[po];
            const float dsum = ((tau & 1u) ? -c4.x : c4.x) * axx + ((tau & 2u) ? -c4.y : c4.y) * axy
                               + ((tau & 4u) ? -c4.z : c4.z) * axz;
            if (!(dsum >= wave_thr)) continue;
            visit(po, (1u << tau) | (dsum >= core_cos ? (256u << tau) : 0u));
        }
    } else {
    // The level's patch octets are dealt round-robin to the runs (run r owns q = r, r + n_runs,
    // ...), so every run sees every region of the sphere and the waves of one output unit carry
    // equal shares of its cone.
        const uint32_t n_subsets = n_runs, subset = run;
        const uint32_t my_count = (n_po + n_subsets - 1 - subset) / n_subsets;

        for (uint32_t base = 0; base < my_count; base += 64) {
            // ---- candidates among the next 64 patch octets of the subset
            const uint32_t qi = base + lane;
            const uint32_t q = qi * n_subsets + subset;
            uint32_t cmask = 0;
            if (qi < my_count) {
                const float4 c4 = pn[q];
                const float a = c4.x * axx, b = c4.y * axy, c = c4.z * axz;
#pragma unroll
                for (int tau = 0; tau < 8; tau++) {
                    float dsum = ((tau & 1) ? -a : a) + ((tau & 2) ? -b : b) + ((tau & 4) ? -c : c);
                    if (dsum >= wave_thr) cmask |= 1u << tau;
                    if (dsum >= core_cos) cmask |= 256u << tau;
                }
                if (MARKS == 1 && c4.w != 0.0f) cmask = 0;
            }
            unsigned long long bal = __ballot((cmask & 255u) != 0u);
            while (bal) {                                  // ascending patch order
                const int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                visit(__builtin_amdgcn_readlane(q, src), __builtin_amdgcn_readlane(cmask, src));
            }
        }
    }
    if (acc_terms != 0u) flush(acc_tau);

    if (MARKS == 2) {
        if (walked_any != 0u && lane_valid) {
#pragma unroll
            for (int k = 0; k < GO_SUMS; k++)
                partial[((size_t)run_local * GO_SUMS + k) * oct_local + lo] += s_run[k][lane];
        }
    } else if (lane_valid) {
#pragma unroll
        for (int k = 0; k < GO_SUMS; k++)
            partial[((size_t)run_local * GO_SUMS + k) * oct_local + lo] = s_run[k][lane];
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            tot_tested += __shfl_xor(tot_tested, off);
            tot_accepted += __shfl_xor(tot_accepted, off);
        }
        if (lane == 0) {
            atomicAdd(&counts[1], tot_tested);
            atomicAdd(&counts[2], tot_accepted);
            atomicAdd(&counts[3], tot_walked);       // records this wave walked (64 lanes each, once per live pattern),
            atomicAdd(&counts[4], tot_evaluated);    // ... and those it evaluated past the cosine
            atomicAdd(&counts[6], tot_visited);      // records of the patch octets it walked, once each
        }
    }
}

// First stage of the run sum when a level has many runs: group g adds its stretch of
// consecutive runs, in order, into one more set of partial sums.  One thread per
// (output octet, member, group).
// (partial holds the n_runs runs of one launch; its groups are group_base, group_base + 1, ...)
// add_to: the group's sums so far are the starting value (a level whose launches all add into one group).
__global__ __launch_bounds__(256) void ggx_oct_reduce(const float* __restrict__ partial, uint32_t n_runs,
                                                       uint32_t per_group, uint32_t group_base, uint32_t n_oct,
                                                       const uint32_t* __restrict__ units, uint32_t oct_count,
                                                       uint32_t add_to, double* __restrict__ reduced) {
    const uint32_t lo = blockIdx.x * GO_THREADS + threadIdx.x;
    if (lo >= oct_count) return;
    const uint32_t gidx = go_global(lo, units);
    const uint32_t m = blockIdx.y, g = blockIdx.z;
    if (gidx >= n_oct) return;
    const uint32_t s0 = g * per_group, s1 = min(n_runs, s0 + per_group);
    // m < 8: the member's RGB sums; m == 8: the octet's weight sum
    const uint32_t j0 = m < 8 ? m * 3 : 24, nj = m < 8 ? 3 : 1;
    double v[3] = {0.0, 0.0, 0.0};
    if (add_to)
        for (uint32_t k = 0; k < nj; k++) v[k] = reduced[((size_t)(group_base + g) * GO_SUMS + (j0 + k)) * oct_count + lo];
    for (uint32_t sl = s0; sl < s1; sl++) {
        for (uint32_t k = 0; k < nj; k++) v[k] += (double)partial[((size_t)sl * GO_SUMS + (j0 + k)) * oct_count + lo];
    }
    for (uint32_t k = 0; k < nj; k++) reduced[((size_t)(group_base + g) * GO_SUMS + (j0 + k)) * oct_count + lo] = v[k];
}

// One thread per (output octet, member): coalesced over the octet index.  T: float (the runs'
// sums as ggx_oct_pairs stores them) or double (the groups' sums of ggx_oct_reduce).
template <typename T>
__global__ __launch_bounds__(256) void ggx_oct_final(const T* __restrict__ partial, uint32_t n_sums,
                                                      uint32_t out_w, uint32_t tw, const uint32_t* __restrict__ units,
                                                      uint32_t n_oct, uint32_t oct_count,
                                                      uint32_t out_begin, uint32_t out_count, float4* __restrict__ out) {
    const uint32_t lo = blockIdx.x * GO_THREADS + threadIdx.x;
    if (lo >= oct_count) return;
    const uint32_t gidx = go_global(lo, units);
    const uint32_t m = blockIdx.y;
    if (gidx >= n_oct) return;
    uint32_t axis, oj, oi;
    go_decode(gidx, out_w / 2, tw, axis, oj, oi);
    Texel t = octet_member(axis, oj, oi, m, out_w, out_w);
    const uint32_t o = (t.face * out_w + t.y) * out_w + t.x;
    if (o - out_begin >= out_count) return;
    double v[3] = {0.0, 0.0, 0.0}, w = 0.0;
    for (uint32_t sl = 0; sl < n_sums; sl++) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] += (double)partial[((size_t)sl * GO_SUMS + (m * 3 + k)) * oct_count + lo];
        w += (double)partial[((size_t)sl * GO_SUMS + 24) * oct_count + lo];
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

uint32_t ggx_oct_tile_edge(uint32_t out_w) { return go_tile_edge(out_w / 2); }

// The runs of a level, the launches they are spread over and the bytes of their sums, for a kernel whose
// waves each own 64 lanes of output (octets, or hexadecets in ggx_hex.hip) with `sums` FP32 sums per lane
// and `members` texels behind a lane and behind a record, `patterns` = members.  pl.lv, pl.n_po and
// pl.matrix are set by the caller; lanes_level and units_level describe the whole level (runs and
// subsets come from the level geometry only: range/shard independence), n_units the call.
void ggx_plan_runs(GgxOctPlan& pl, uint32_t lanes_level, uint32_t units_level, uint32_t members, uint32_t sums,
                   size_t rec_each, uint32_t n_units, double max_waves) {
    pl.accumulate = false;
    const uint32_t ps = pl.lv.patch;
    // Enough waves that an eighth of the level (one of 8 GPUs) still fills a GPU (6144 wave
    // slots each) a few times over: 65536.  Large levels get up to sixteen times that, as far
    // as their work can carry the partial sums (every (wave, run) writes 64 x `sums` floats; that
    // traffic is kept to a few per cent of the estimated run time).
    const double cmin = fmax(0.0, 2.0 * (double)pl.lv.cutoff - 1.0);
    const double cone = 0.5 * (1.0 - cmin);                              // accepted share of the sphere
    const double wave_records = (double)lanes_level / 64.0 * (double)pl.n_po * ps * ps * (double)members * cone;
    const double seconds = wave_records * 70e-9 / 1024.0;                // ~70 ns per wave, record and pattern per SIMD
    double waves = 0.04 * seconds * 1.5e12 / (64.0 * sums * 4.0);
    waves = fmin(fmax(waves, 65536.0), max_waves);
    if (tuning().ggx_waves > 0.0) waves = tuning().ggx_waves;            // tuning knob
    uint32_t want = (uint32_t)ceil(waves / (double)units_level);
    // The waves in flight on an XCD work on one run (see ggx_oct_pairs) and on neighbouring
    // output units; together they read a good part of that run's records, each many times, and
    // that set should sit in the XCD's 4 MiB L2.  Measured on the 2048 -> 1024 level with the
    // runs dealt to the XCDs: 2 and 3.7 MiB of records per run 755 ms, 7.3 MiB 849 ms, 14.7 MiB
    // 1004 ms.  So at most 4 MiB of records per run; the partial sums that costs are kept below
    // 4 GiB by launching the runs in several passes (below).
    {
        const double rec_total = (double)pl.n_po * ps * ps * (double)rec_each;
        const double per_run = (double)units_level * GO_UNIT * sums * sizeof(float);
        const double fit = ceil(rec_total / (double)(4u << 20));
        // ... and the partial sums should not be many times the bytes the level has to move anyway
        // (32 per input texel, 16 per output texel): at most 8 times those.
        const double algorithmic = 32.0 * pl.n_po * ps * ps * (double)members + 16.0 * lanes_level * (double)members;
        const double modest = floor(8.0 * algorithmic / per_run);
        if (tuning().ggx_waves <= 0.0) {
            if ((double)want > modest) want = modest >= 1.0 ? (uint32_t)modest : 1u;
            if (fit > (double)want) want = (uint32_t)fit;
        }
        uint32_t max_runs = pl.n_po < 32000u ? pl.n_po : 32000u;     // at least one patch per run; the run numbers travel in 15 bits
        if (max_runs == 0) max_runs = 1;
        uint32_t runs = want < 1 ? 1 : (want > max_runs ? max_runs : want);
        // The sums of one launch's runs (FP32) and of the groups of runs (FP64) together stay below
        // 4 GiB for the whole level.  What that cannot hold at once is launched in several passes
        // over the same buffer, each followed by its groups' sums (2048 -> 1024: 120 runs as 3 x 40,
        // 4.1 GB; the same bits as one launch would give).  A group never spans two launches, and a
        // launch has a multiple of 8 runs, so that the 8 XCDs can own equal shares of them (see
        // ggx_oct_pairs).  Levels too large even for 8 runs at a time (8192 -> 4096: 1.3 GB per run)
        // launch what fits and add every launch into one set of FP64 sums (below).
        const double elems = (double)units_level * GO_UNIT * sums;
        const double budget = tuning().ggx_sums_mib > 0.0 ? tuning().ggx_sums_mib * 1048576.0 : (double)(4ull << 30);
        auto groups_of = [](uint32_t n) { uint32_t g = n > 64 ? (n + 31) / 32 : 1; return g > 64 ? 64u : g; };
        auto single_fits = [&](uint32_t n) {
            const uint32_t g = groups_of(n);
            return elems * ((double)n * sizeof(float) + (g > 1 ? (double)g * sizeof(double) : 0.0)) <= budget;
        };
        pl.passes = 0;
        if (single_fits(runs)) {
            pl.passes = 1;
            pl.runs_per_pass = runs >= 16 ? runs / 8 * 8 : runs;
        } else {
            for (uint32_t passes = 2; passes <= 64 && pl.passes == 0; passes++) {
                const uint32_t per_pass = ((runs + passes - 1) / passes + 7) / 8 * 8;
                uint32_t per_group = 8;
                for (uint32_t g = 32; g >= 8; g--)
                    if (per_pass % g == 0) { per_group = g; break; }
                const uint32_t groups = passes * (per_pass / per_group);
                if (elems * ((double)per_pass * sizeof(float) + (double)groups * sizeof(double)) <= budget) {
                    pl.passes = passes;
                    pl.runs_per_pass = per_pass;
                    pl.per_group = per_group;
                }
            }
            if (pl.passes == 0) {
                // Not even 8 runs at a time (8192 -> 4096: 1.3 GB per run).  The few runs that fit one launch
                // would each carry thousands of FP32 chunks in their running sums, and small chunks get lost
                // in a large running sum (4096 -> 2048 forced to 1 / 3 / 8 runs: 1.8e-5 / 5e-6 / 1.2e-6 against
                // the CPU statement).  So: up to 32 runs, launched as many at a time as fit beside ONE set of
                // FP64 sums, which every launch adds its runs into.
                uint32_t total = runs < 32u ? runs : 32u;
                uint32_t r = total;
                while (r > 1 && elems * ((double)r * sizeof(float) + sizeof(double)) > budget) r--;
                if (r >= 8) r = r / 8 * 8;
                pl.accumulate = true;
                pl.runs_per_pass = r;
                pl.passes = (total + r - 1) / r;
                pl.per_group = r;
            }
        }
        pl.n_runs = pl.runs_per_pass * pl.passes;
    }
    pl.out_units = n_units;
    const uint32_t oct_count = pl.out_units * GO_UNIT;
    // many runs are summed in two stages (fixed grouping, so still one summation order)
    if (pl.accumulate) {
        pl.reduce_groups = 1;
    } else if (pl.passes > 1) {
        pl.reduce_groups = pl.n_runs / pl.per_group;
    } else {
        pl.reduce_groups = pl.n_runs > 64 ? (pl.n_runs + 31) / 32 : 1;
        if (pl.reduce_groups > 64) pl.reduce_groups = 64;
        pl.per_group = (pl.n_runs + pl.reduce_groups - 1) / pl.reduce_groups;
    }
    const uint32_t stage2 = pl.accumulate ? 1 : (pl.reduce_groups > 1 ? pl.reduce_groups : 0);
    pl.partial_bytes = (size_t)sums * oct_count * (pl.runs_per_pass * sizeof(float) + stage2 * sizeof(double));
}

GgxOctPlan ggx_oct_plan(uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips, uint32_t n_units) {
    GgxOctPlan pl;
    pl.hex = false;
    pl.lv = ggx_level_params(mip, mips, in_w, in_w);
    const uint32_t ps = pl.lv.patch, pw = in_w / ps;
    pl.n_po = 3 * (pw / 2) * (pw / 2);
    pl.tw = go_tile_edge(out_w / 2);
    const uint32_t n_oct_level = 3 * (out_w / 2) * (out_w / 2);
    const uint32_t units_level = (n_oct_level + GO_UNIT - 1) / GO_UNIT;      // one wave each per run
    // The matrix form takes whole groups of 16 records (inputs of 64 x 64 and more).  It cannot drop
    // single records after their cosine as EARLY does at wide lobes, and is still the faster one there
    // (2048 -> 256: 323 ms against 462; 512 -> 64: 1.64 against 2.16).
    pl.matrix = tuning().ggx_matrix != 0 && (ps * ps) % GO_MX_GROUP == 0;
    const size_t rec_each = pl.matrix ? (size_t)GO_MX_REC_BYTES : GO_REC * sizeof(float4);
    ggx_plan_runs(pl, n_oct_level, units_level, 8, GO_SUMS, rec_each, n_units);
    // (the counting variant walks the vector form's records: room for the larger of the two)
    // Matrix form with marked patch octets (ggx_oct_mark_patches): the matrix form's records, then room for
    // the vector form's, of which only the marked patch octets' are written and read.
    pl.marks = pl.matrix && tuning().ggx_few > 0.0f;
    pl.mx_bytes = pl.matrix ? (size_t)pl.n_po * ps * ps * (size_t)GO_MX_REC_BYTES : 0;
    pl.rec_bytes = pl.mx_bytes + (!pl.matrix || pl.marks ? (size_t)pl.n_po * ps * ps * GO_REC * sizeof(float4) : 0);
    pl.pn_bytes = ((size_t)pl.n_po + 1) * sizeof(float4) + (size_t)pl.n_po * sizeof(uint32_t);    // + number and list of the marked patch octets
    return pl;
}

hipError_t ggx_oct_launch(const GgxOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_w,
                          const uint32_t* d_units, uint32_t out_begin, uint32_t out_count,
                          float* d_out, void* d_rec, void* d_pn, double* d_partial, unsigned long long* d_counts,
                          hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t ps = pl.lv.patch, ps2 = ps * ps, pw = in_w / ps;
    const uint32_t n_rec = pl.n_po * ps2;
    const bool matrix = pl.matrix && !d_counts;
    const bool marks = matrix && pl.marks;
    float4* const d_vec = (float4*)((char*)d_rec + pl.mx_bytes);       // the vector form's records beside the matrix form's
    ggx_oct_prep_patches<<<(pl.n_po + 255) / 256, 256, 0, stream>>>(pw, pl.n_po, (float4*)d_pn);
    if (marks) {
        // a patch of fewer than 2 n_min texels cannot hold n_min equal ones: half its texels then
        // (a knob value above 256, more than any patch holds, marks every patch: tests)
        const float n_min = tuning().ggx_few > 256.0f ? tuning().ggx_few : fminf(tuning().ggx_few, 0.5f * (float)ps2);
        ggx_oct_mark_patches<<<pl.n_po, 256, 0, stream>>>((const float4*)d_env, in_w, ps, pl.n_po, n_min, (float4*)d_pn);
        ggx_oct_list_marks<<<1, 1024, 0, stream>>>(pl.n_po, (float4*)d_pn);
    }
    if (matrix)
        ggx_mx_prep<<<(n_rec / GO_MX_GROUP * 64 + 255) / 256, 256, 0, stream>>>(
            (const float4*)d_env, in_w, ps, n_rec, (float4*)d_rec, (go_u32x4*)((float4*)d_rec + n_rec));
    if (marks)
        ggx_oct_prep<<<(n_rec + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, ps, n_rec, (const float4*)d_pn, d_vec);
    else if (!matrix)
        ggx_oct_prep<<<(n_rec + 255) / 256, 256, 0, stream>>>((const float4*)d_env, in_w, ps, n_rec, nullptr, (float4*)d_rec);
    const double cut = (double)pl.lv.cutoff;
    const float cmin = (float)fmax(0.0, 2.0 * cut - 1.0);
    const float theta_acc = (float)acos(fmin(1.0, fmax(-1.0, 2.0 * cut - 1.0)));
    const float a2m1 = pl.lv.alpha2 - 1.0f;
    const float core_thr = ggx_core_threshold(pl.lv.alpha2);
    // One wave per (unit, run).  Runs are dealt to the XCDs (see ggx_oct_pairs) when several
    // runs would otherwise share each L2: the call covers fewer units than the GPU has wave slots
    // (a shard: an eighth of the 2048 -> 512 level 231 -> 181 ms, of 2048 -> 256 75 -> 62 ms), the
    // level is in the narrow-lobe regime, where a record feeds one weight evaluation and its load
    // latency shows, or a run's records are more than half an L2 (2048 -> 1024 level, an eighth of
    // the units: 150 -> 127 ms).  Whole wide-lobe levels with small runs lose 1-3 % with it.  The
    // order changes no sum.
    const double cone = 0.5 * (1.0 - fmax(0.0, 2.0 * (double)pl.lv.cutoff - 1.0));
    const double run_bytes = (double)(pl.matrix ? pl.mx_bytes : pl.rec_bytes) / (double)pl.n_runs;
    uint32_t xcd_runs = (pl.runs_per_pass % 8 == 0 &&
                         (pl.out_units < 3072u || 8.0 * cone < 1.5 || run_bytes > (double)(2u << 20))) ? 1u : 0u;
    if (tuning().ggx_xcd_slices >= 0) xcd_runs = (tuning().ggx_xcd_slices != 0 && pl.runs_per_pass % 8 == 0) ? 1u : 0u;
    const dim3 grid(pl.runs_per_pass * pl.out_units);
    const uint32_t small_blocks = (pl.out_units * GO_UNIT + GO_THREADS - 1) / GO_THREADS;
    const uint32_t n_oct = 3 * (out_w / 2) * (out_w / 2);
    const uint32_t oct_count = pl.out_units * GO_UNIT;
    // workspace: the groups' sums (double) first, then the sums of one launch's runs (float)
    double* d_groups = d_partial;
    const bool two_stage = pl.accumulate || pl.reduce_groups > 1;
    float* d_runs = (float*)(d_partial + (size_t)(two_stage ? pl.reduce_groups : 0) * GO_SUMS * oct_count);
    const bool early = cmin == 0.0f && tuning().ggx_early != 0;      // the window reaches the horizon
    // At those levels the reference's patch test lets patches through up to acos(2 cutoff - 1) > 90
    // degrees from an output direction, and then rejects every one of their texels below the
    // horizon (cube.cpp:271): a third to a half of the texels it tests.  A patch octet further than
    // 90 degrees + its own angular radius from every direction of a wave has weight 0 throughout,
    // so the wave's bounding-cone test leaves it out (the counting variant keeps the reference's
    // angle: its counts are the reference's).
    const float theta_walk = early ? fminf(theta_acc, 1.5707964f + 1.4143f / (float)pw) : theta_acc;
    // The per-record test (EARLY) makes every record wait for a vector compare to reach a scalar
    // branch: worth it only while other waves fill that time (512 -> 16: 2304 waves, 0.37 -> 0.54 ms
    // with it).  It changes no sum -- a skipped record has weight 0 in every lane -- so the choice
    // may depend on the call's size.
    const bool early_records = early && !matrix && (uint64_t)pl.runs_per_pass * pl.out_units >= 24576u;
    // The window ends above the horizon (with room for the rounding of t): the weights may take the
    // v_rsq_f32 form of go_weight_quad.  Measured: 2048 -> 1024 511 -> 495 ms, 2048 -> 512 902 -> 872 ms,
    // 1024 -> 512 46.8 -> 46.2 ms, but 512 -> 256 5.9 -> 6.4 ms; so from 1024 x 1024 inputs on
    // (knob ggx_rsq: 0 never, 2 wherever the window allows).
    const bool narrow = pl.lv.cutoff >= 0.51f && (tuning().ggx_rsq == 2 || (tuning().ggx_rsq == 1 && in_w >= 1024u));
    if (ev0) (void)hipEventRecord(ev0, stream);
    for (uint32_t pass = 0; pass < pl.passes; pass++) {
        const uint32_t run_base = pass * pl.runs_per_pass;
        const uint32_t packed = pl.n_runs | (run_base << 15) | (xcd_runs << 31);
        auto go_on = [&](auto kernel, const float4* records) {
            kernel<<<grid, GO_WG, 0, stream>>>(records, (const float4*)d_pn, d_units, pl.n_po,
                ps2, out_w, pl.tw, n_oct, oct_count, packed, out_begin, out_count,
                pl.lv.cutoff, a2m1, cmin, d_counts ? theta_acc : theta_walk, core_thr, d_runs, d_counts);
        };
        auto go = [&](auto kernel) { go_on(kernel, (const float4*)d_rec); };
        if (d_counts) { if (early_records) go(ggx_oct_pairs<true, true>); else go(ggx_oct_pairs<true, false>); }
        else if (marks) {
            // all but the marked patch octets on the matrix pipe, then the marked ones with FP32 products, added
            // to the same sums by the same waves (its waves leave at once when nothing is marked)
            if (narrow) go(ggx_oct_pairs<false, false, true, true, 1>); else go(ggx_oct_pairs<false, false, true, false, 1>);
            go_on(ggx_oct_pairs<false, false, false, false, 2>, d_vec);
        }
        else if (matrix) { if (narrow) go(ggx_oct_pairs<false, false, true, true>); else go(ggx_oct_pairs<false, false, true, false>); }
        else          { if (early_records) go(ggx_oct_pairs<false, true>); else go(ggx_oct_pairs<false, false>); }
        if (pass + 1 == pl.passes && ev1) (void)hipEventRecord(ev1, stream);
        if (pl.accumulate)
            ggx_oct_reduce<<<dim3(small_blocks, 9, 1), GO_THREADS, 0, stream>>>(
                d_runs, pl.runs_per_pass, pl.runs_per_pass, 0, n_oct, d_units, oct_count, pass > 0 ? 1u : 0u, d_groups);
        else if (pl.reduce_groups > 1)
            ggx_oct_reduce<<<dim3(small_blocks, 9, pl.passes > 1 ? pl.runs_per_pass / pl.per_group : pl.reduce_groups),
                             GO_THREADS, 0, stream>>>(d_runs, pl.runs_per_pass, pl.per_group, run_base / pl.per_group, n_oct,
                                                      d_units, oct_count, 0u, d_groups);
    }
    if (two_stage)
        ggx_oct_final<double><<<dim3(small_blocks, 8), GO_THREADS, 0, stream>>>(
            d_groups, pl.reduce_groups, out_w, pl.tw, d_units, n_oct, oct_count, out_begin, out_count, (float4*)d_out);
    else
        ggx_oct_final<float><<<dim3(small_blocks, 8), GO_THREADS, 0, stream>>>(
            d_runs, pl.n_runs, out_w, pl.tw, d_units, n_oct, oct_count, out_begin, out_count, (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
