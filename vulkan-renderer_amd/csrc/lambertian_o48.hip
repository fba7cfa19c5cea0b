// Lambertian irradiance through all 48 symmetries of the cube, with the clamped cosine split into
// its even and odd halves (gfx950).
//
// lambertian_hex.hip shares work between the 16 texels that the sign flips and the exchange of x and y
// make of one: those leave the reference's dot product ((0 + a.x b.x) + a.y b.y) + a.z b.z
// (linear.hpp:100-107) unchanged to the bit.  The other axis exchanges move a product out of the inner
// sum, so a texel's cosine may differ from the reference's in its last bit -- which the split form can
// afford: it is not bit-exact anyway, it is held to ~1e-6 by its own cancellation mark (lo_final), and a
// last-bit difference of either sign in single cosines averages out over the input.  So this kernel uses
// the whole group G of the cube: the 8 sign flips D_t and the 6 axis permutations P (48 elements).
//
// An ORBIT is the set of texels a texel's direction reaches under G: 48 texels, 24 on the face diagonals.
// Canonical member: (c_i, c_j, 1) on the +Z face with i <= j (c_i: the i-th positive texel-centre
// coordinate); member (t, p) has the direction D_t P_p n0, (P_p n0)_x = n0[PERM[p][x]].  For an output
// orbit with canonical direction n0 and an input orbit with canonical weight vector g0 (the direction over
// the fourth power of its length, cube.cpp:116-117) the cosine of members (t, r) and (s, p) is
//     v_tau(q) = q_x n0_x g0[tau(x)] + q_y n0_y g0[tau(y)] + q_z n0_z g0[tau(z)]
// with tau = p o r^-1 and q the signs ts carried back through r.  The even half |v| of max(0, v) summed
// over an orbit pair is therefore a convolution over G; |v| does not see the common sign, which leaves the
// group of order 24 (the rotations of the cube, S4), and its Fourier transform turns the 48 x 48 = 2304
// texel pairs into 64 products per colour channel:
//   * over the sign flips a Walsh-Hadamard transform (even characters k = 0, xy, xz, yz), written
//     as max / min / median of a + b, a - b, c as in lambertian_split.hip: w_tau[k], L_p[k];
//   * k = 0: a convolution over the 6 permutations: the trivial and the sign representation (one product
//     each) and the two-dimensional one in integer matrices (a 2 x 2 matrix product: 8);
//   * k = xy, xz, yz are permuted among themselves; sums and differences over the two permutations
//     (one even, one odd) that carry k to k' split them into two 3 x 3 matrix products (27 + 27).
// The sums and differences ride in the two halves of packed instructions, so a record costs 96 packed
// multiply-adds and 76 other vector instructions for 2304 texel pairs (lambertian_hex.hip: 37 for 256).
// The linear (odd) half of the clamped cosine is the first moment of the input, as in lambertian_split.hip;
// the marks for texels that cancel too far and their second pass are that file's and lambertian_octets.hip's.
//
// Kernel plan
//   lo_prep     a workgroup per 64 input orbits, a wave per colour channel: the orbits' transformed radiance
//               (72 floats) and weight vectors, and their share of the first moment.
//   lo_pairs    grid = (output units of 64 orbits) x (groups of 4 input slices); a wave per slice walks its
//               records with scalar loads: the next record's 72 words are fetched block by block as this
//               record's multiply-adds let go of them, while the vector pipe already prepares the next
//               record's weights.  FP32 sums over a chunk of <= 320 records; at the end of a chunk the four
//               waves' sums meet in LDS and each wave adds 18 of the 72 in FP64; a group's sums leave as FP32 (one
//               rounding of 2^-25 each, hundreds of groups: nothing next to the FP32 sums they come from).
//   lo_reduce   all groups of a unit in FP64; one workgroup adds the first moment's parts.
//   lo_final    inverse transforms, the odd half, the scale, the cancellation measured
//               per texel; marks are set per OCTET in the numbering of lambertian_octets.hip, whose kernels
//               redo the marked octets.
#include "ibl_device.h"
#include "ibl_kernels.h"

namespace ibl {

constexpr int LO_RUN = 4;                   // slices (waves) per workgroup
constexpr int LO_THREADS = 64 * LO_RUN;
constexpr int LO_RECF = 72;                 // floats per record
constexpr int LO_SUMS = 72;
constexpr int LO_OWN = LO_SUMS / LO_RUN;    // sums whose FP64 totals one wave keeps
constexpr uint32_t LO_CHUNK_MAX = 320;      // records per FP32 sum at most
constexpr uint32_t LO_OCT_BLOCK = 256;      // the octet blocks of lambertian_octets.hip (marks)
constexpr int LO_PREP_THREADS = 192;      // 64 orbits x 3 colour channels

// The permutations (P_p v)_x = v[PERM[p][x]], their parities and inverses; the even and the odd permutation
// that carry character a to character b (characters 0, 1, 2 = xy, xz, yz); the two-dimensional representation
// of S3 on the plane x + y + z = 0 in the basis (e_x - e_y, e_y - e_z), R2I[p] = R2(P_p^-1).
#define LO_TABLES                                                                                        \
    const int PERM[6][3] = { {0, 1, 2}, {1, 0, 2}, {2, 1, 0}, {2, 0, 1}, {0, 2, 1}, {1, 2, 0} };          \
    const int PAR[6] = { 0, 1, 1, 0, 1, 0 };                                                              \
    const int INV[6] = { 0, 1, 2, 5, 4, 3 };                                                              \
    const int EVEN[3][3] = { {0, 3, 5}, {5, 0, 3}, {3, 5, 0} };                                           \
    const int ODD[3][3] = { {1, 4, 2}, {4, 2, 1}, {2, 1, 4} };                                            \
    const int R2I[6][2][2] = { {{1, 0}, {0, 1}}, {{-1, 1}, {0, 1}}, {{0, -1}, {-1, 0}},                   \
                               {{0, -1}, {1, -1}}, {{1, 0}, {1, -1}}, {{-1, 1}, {-1, 0}} };               \
    (void)PERM; (void)PAR; (void)INV; (void)EVEN; (void)ODD; (void)R2I;

__device__ __forceinline__ double lo_sign(uint32_t k, uint32_t s) { return (__popc(k & s) & 1) ? -1.0 : 1.0; }

// Record of an input orbit (72 floats; ch = colour channel):
//   [2 ch], [2 ch + 1]                 k = 0: sum over the permutations, sum with their parities
//   [6 + 6 l + 2 ch + j]               k = 0: sum_p L_p R2(P_p)[l][j]   (row l of all channels together: the multiply-adds
//                                      of one row then go to six different sums, and no one waits for the one before)
//   [18 + 2 (9 c + 3 ch + a)], + 1     k = a: L_even(a -> c) + L_odd(a -> c), L_even - L_odd   (c-major over all channels: a
//                                      sum is then touched again 27 multiply-adds later, not three; after three or nine the
//                                      assembler puts a wait state in front of the next one, an issue slot each)
// and in a separate array the weight vector (g.x, g.y, g.z, g.z).  A workgroup is 64 orbits, a wave per colour
// channel (24 FP64 sums a thread; one thread for all three channels holds 350 registers and the chip a wave per
// SIMD).  (The kernel also clears the call's marks.)
__global__ __launch_bounds__(LO_PREP_THREADS) void lo_prep(const float4* __restrict__ env, uint32_t w, uint32_t n_rec,
                                                          float* __restrict__ rec, float4* __restrict__ recg,
                                                          double* __restrict__ moment_parts,
                                                          uint32_t* __restrict__ flags, uint32_t n_flags) {
    LO_TABLES
    __shared__ float s_rec[64 * (LO_RECF + 1)];
    __shared__ double s_m[9][64];
    const uint32_t tid = threadIdx.x, ql = tid & 63u;
    const int ch = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t q = blockIdx.x * 64u + ql;
    for (uint32_t i = blockIdx.x * blockDim.x + tid; i < n_flags; i += gridDim.x * blockDim.x) flags[i] = 0u;
    double m[3] = { 0.0, 0.0, 0.0 };
    double lt = 0.0, ls = 0.0, lr[2][2] = { {0.0, 0.0}, {0.0, 0.0} }, la[3][3][2];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) la[a][cc][0] = la[a][cc][1] = 0.0;
    const O48Lane ol = o48_tiled_lane(q, w / 2);                  // (input orbits are numbered tile by tile; the rest get a zero record)
    if (ol.valid) {
        Texel c = octet_member(2, ol.j, ol.i, 0, w, w);
        F3 d = ref_normal_from(c.face, c.x, c.y, w, w);           // (c_i, c_j, 1)
        float l2 = ref_dot(d, d);
        float inv = 1.0f / (l2 * l2);                             // cube.cpp:116-117
        const float g[3] = { d.x * inv, d.y * inv, d.z * inv };
        if (ch == 0) recg[q] = make_float4(g[0], g[1], g[2], g[2]);
        const uint32_t ks[4] = { 0u, 3u, 5u, 6u };
#pragma unroll
        for (int p = 0; p < 6; p++) {
            float lv[8];
#pragma unroll
            for (uint32_t s = 0; s < 8; s++) {
                lv[s] = 0.0f;
                if (!(ol.diagonal && (p & 1))) {                  // (on a diagonal the odd-numbered members are the even-numbered ones)
                    Texel t = o48_member(ol.j, ol.i, (uint32_t)p * 8u + s, w);
                    lv[s] = ((const float*)env)[(size_t)((t.face * w + t.y) * w + t.x) * 4u + (uint32_t)ch];      // cube.cpp:104 row stride (square faces)
                }
            }
            double lh[4];
#pragma unroll
            for (int kq = 0; kq < 4; kq++) {
                double v = 0.0;
#pragma unroll
                for (uint32_t s = 0; s < 8; s++) v += lo_sign(ks[kq], s) * (double)lv[s];
                lh[kq] = v;
            }
            lt += lh[0];
            ls += PAR[p] ? -lh[0] : lh[0];
#pragma unroll
            for (int l = 0; l < 2; l++)
#pragma unroll
                for (int j = 0; j < 2; j++) lr[l][j] += (double)R2I[INV[p]][l][j] * lh[0];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    if (EVEN[a][cc] == p) { la[a][cc][0] += lh[a + 1]; la[a][cc][1] += lh[a + 1]; }
                    if (ODD[a][cc] == p) { la[a][cc][0] += lh[a + 1]; la[a][cc][1] -= lh[a + 1]; }
                }
            // first moment: member (s, p) points to (+-g[PERM[p][x]], ...)
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                double v = 0.0;
#pragma unroll
                for (uint32_t s = 0; s < 8; s++) v += lo_sign(1u << ax, s) * (double)lv[s];
                m[ax] += (double)g[PERM[p][ax]] * v;
            }
        }
    }
    if (!ol.valid && ch == 0 && q < n_rec) recg[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // The workgroup's records leave through LDS, so that neighbouring threads store neighbouring words
    // (rows of 73 words keep the LDS writes off each other's banks).
    {
        float* row = s_rec + ql * (LO_RECF + 1);
        row[2 * ch] = (float)lt;
        row[2 * ch + 1] = (float)ls;
#pragma unroll
        for (int l = 0; l < 2; l++)
#pragma unroll
            for (int j = 0; j < 2; j++) row[6 + 6 * l + 2 * ch + j] = (float)lr[l][j];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                row[18 + 2 * (9 * cc + 3 * ch + a)] = (float)la[a][cc][0];
                row[18 + 2 * (9 * cc + 3 * ch + a) + 1] = (float)la[a][cc][1];
            }
#pragma unroll
        for (int ax = 0; ax < 3; ax++) s_m[ax * 3 + ch][ql] = m[ax];
        __syncthreads();
        const size_t base = (size_t)blockIdx.x * 64 * LO_RECF, end = (size_t)n_rec * LO_RECF;
        for (uint32_t f = tid; f < (uint32_t)(64 * LO_RECF); f += LO_PREP_THREADS)
            if (base + f < end) rec[base + f] = s_rec[(f / LO_RECF) * (LO_RECF + 1) + f % LO_RECF];
    }
    for (uint32_t step = 32; step > 0; step >>= 1) {      // fixed tree
        if (tid < step) {
#pragma unroll
            for (int k = 0; k < 9; k++) s_m[k][tid] += s_m[k][tid + step];
        }
        __syncthreads();
    }
    if (tid < 9) moment_parts[(size_t)blockIdx.x * 9 + tid] = s_m[tid][0];
}

// ---- the pair kernel ------------------------------------------------------------------------------------

typedef float lo_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lo_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lo_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t lo_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t lo_u32x16 __attribute__((ext_vector_type(16)));

// A record in scalar registers: four blocks of 16 words and one of 8.  Loads and the wait are written out
// (the compiler would sink a load to its first use); every block is fetched again, for the next record, as
// soon as this record's last multiply-add that reads it has been issued.
struct LoRec { lo_u32x16 b0, b1, b2, b3; lo_u32x8 b4; };

template <int BLOCK>
__device__ __forceinline__ void lo_block_load(LoRec& r, const float* __restrict__ p) {
    if constexpr (BLOCK == 0) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(r.b0) : "s"(p));
    else if constexpr (BLOCK == 1) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=&s"(r.b1) : "s"(p));
    else if constexpr (BLOCK == 2) asm volatile("s_load_dwordx16 %0, %1, 0x80" : "=&s"(r.b2) : "s"(p));
    else if constexpr (BLOCK == 3) asm volatile("s_load_dwordx16 %0, %1, 0xc0" : "=&s"(r.b3) : "s"(p));
    else asm volatile("s_load_dwordx8 %0, %1, 0x100" : "=&s"(r.b4) : "s"(p));
}
__device__ __forceinline__ void lo_g_load(lo_u32x4& g, const float4* __restrict__ p) {
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(g) : "s"(p));
}
__device__ __forceinline__ void lo_wait(LoRec& r, lo_u32x4& ga, lo_u32x4& gb) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.b0), "+s"(r.b1), "+s"(r.b2), "+s"(r.b3), "+s"(r.b4), "+s"(ga), "+s"(gb));
}
// Words 2 PAIR, 2 PAIR + 1 of the record.
template <int PAIR>
__device__ __forceinline__ lo_u32x2 lo_pair(const LoRec& r) {
    constexpr int at = 2 * (PAIR % 8);
    if constexpr (PAIR / 8 == 0) return lo_u32x2{r.b0[at], r.b0[at + 1]};
    else if constexpr (PAIR / 8 == 1) return lo_u32x2{r.b1[at], r.b1[at + 1]};
    else if constexpr (PAIR / 8 == 2) return lo_u32x2{r.b2[at], r.b2[at + 1]};
    else if constexpr (PAIR / 8 == 3) return lo_u32x2{r.b3[at], r.b3[at + 1]};
    else return lo_u32x2{r.b4[at], r.b4[at + 1]};
}

// (x.SA + y.SB, x.SA - y.SB): sum and difference in the two halves
template <int SA, int SB>
__device__ __forceinline__ lo_f32x2 lo_addsub(lo_f32x2 x, lo_f32x2 y) {
    lo_f32x2 o;
    if constexpr (SA == 0 && SB == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(o) : "v"(x), "v"(y));
    else if constexpr (SA == 0 && SB == 1) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(x), "v"(y));
    else if constexpr (SA == 1 && SB == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(o) : "v"(x), "v"(y));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(o) : "v"(x), "v"(y));
    return o;
}

// What the vector pipe prepares of a record before its multiply-adds: the transformed weights.
struct LoWeights {
    lo_f32x2 ts;          // k = 0: (sum over the permutations, sum with parities)
    lo_f32x2 f2[2];       // k = 0: rows of sum_tau w_tau R2(tau^-1)
    lo_f32x2 wa[3][3];    // [c][b]: (w_even(b -> c) + w_odd(b -> c), w_even - w_odd) at character b
};

// The multiply-adds that read pair PAIR of the record (volatile: they stay in the order written, ahead of the
// load that fetches their block again).  acc: see lo_pairs.
template <int PAIR>
__device__ __forceinline__ void lo_fma_pair(lo_f32x2 (&acc)[LO_SUMS / 2], const LoRec& r, const LoWeights& W) {
    const lo_u32x2 sp = lo_pair<PAIR>(r);
    if constexpr (PAIR < 3) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[PAIR]) : "v"(W.ts), "s"(sp));
    } else if constexpr (PAIR < 9) {
        // row i of the product += f2[i][l] x (row l of the record's matrix)
        constexpr int l = (PAIR - 3) / 3, ch = (PAIR - 3) % 3;
        if constexpr (l == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[3 + 2 * ch]) : "v"(W.f2[0]), "s"(sp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[3 + 2 * ch + 1]) : "v"(W.f2[1]), "s"(sp));
        } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[3 + 2 * ch]) : "v"(W.f2[0]), "s"(sp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[3 + 2 * ch + 1]) : "v"(W.f2[1]), "s"(sp));
        }
    } else {
        constexpr int cc = (PAIR - 9) / 9, ch = ((PAIR - 9) % 9) / 3, a = (PAIR - 9) % 3;
#pragma unroll
        for (int b = 0; b < 3; b++)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[9 + 9 * ch + 3 * a + b]) : "v"(W.wa[cc][b]), "s"(sp));
    }
}
template <int FIRST, int LAST>
__device__ __forceinline__ void lo_fma_range(lo_f32x2 (&acc)[LO_SUMS / 2], const LoRec& r, const LoWeights& W) {
    if constexpr (FIRST < LAST) {
        lo_fma_pair<FIRST>(acc, r, W);
        lo_fma_range<FIRST + 1, LAST>(acc, r, W);
    }
}
// A block's multiply-adds, then its load for the next record.
template <int BLOCK>
__device__ __forceinline__ void lo_fma_block(lo_f32x2 (&acc)[LO_SUMS / 2], LoRec& r, const LoWeights& W, const float* __restrict__ next) {
    lo_fma_range<8 * BLOCK, (BLOCK == 4 ? LO_SUMS / 2 : 8 * BLOCK + 8)>(acc, r, W);
    lo_block_load<BLOCK>(r, next);
}
__device__ __forceinline__ void lo_fma_record(lo_f32x2 (&acc)[LO_SUMS / 2], LoRec& r, const LoWeights& W, const float* __restrict__ next) {
    lo_fma_block<0>(acc, r, W, next);
    lo_fma_block<1>(acc, r, W, next);
    lo_fma_block<2>(acc, r, W, next);
    lo_fma_block<3>(acc, r, W, next);
    lo_fma_block<4>(acc, r, W, next);
}

#ifndef LO_MIN_WAVES
#define LO_MIN_WAVES 2
#endif
#ifndef LO_PREFETCH
#define LO_PREFETCH 1          // the waves touch records ahead of their scalar loads (see lambertian_split.hip); 0: they do not
#endif

// Does the orbit hold a texel of the (face, y, x) range?  (Whole cubes and shards skip the test.)
__device__ __forceinline__ bool lo_lane_wanted(const O48Lane& ol, uint32_t out_size, uint32_t out_begin, uint32_t out_count) {
    if (!ol.valid) return false;
    if (out_begin == 0u && out_count >= 6u * out_size * out_size) return true;
    bool wanted = false;
    for (uint32_t m = 0; m < 48u; m++) {
        Texel t = o48_member(ol.j, ol.i, m, out_size);
        const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
        wanted = wanted || (o - out_begin < out_count);
    }
    return wanted;
}

// COUNT: the wavefronts also count the records they walk.  MULTI: a slice is more than one chunk, so the waves
// keep FP64 totals between the chunks (36 registers that the single-chunk form -- inputs up to 1024 x 1024 faces --
// does without: 4 waves per SIMD instead of 3).
template <bool COUNT, bool MULTI>
__global__ __launch_bounds__(LO_THREADS, LO_MIN_WAVES) void lo_pairs(
    const float* __restrict__ rec, const float4* __restrict__ recg, uint32_t n_rec_in, uint32_t slice_len, uint32_t n_slices,
    uint32_t chunk_len, uint32_t n_chunks, uint32_t out_size, const uint32_t* __restrict__ units, uint32_t n_units,
    uint32_t xcd_affine, uint32_t out_begin, uint32_t out_count, float* __restrict__ partial,
    unsigned long long* __restrict__ counts) {
    // The workgroup: one output unit, LO_RUN consecutive slices, a wave each.
    __shared__ float s_c[LO_RUN][LO_SUMS / 2][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: the record addresses hang on it)
    const uint32_t n_sgroups = (n_slices + LO_RUN - 1) / LO_RUN;
    uint32_t sg, u;
    if (xcd_affine) {                                   // XCD k works on the slice groups k, k + 8, ...
        const uint32_t seq = blockIdx.x >> 3;
        sg = (blockIdx.x & 7u) + 8u * (seq / n_units);
        u = seq % n_units;
        if (sg >= n_sgroups) return;
    } else {
        sg = blockIdx.x / n_units;
        u = blockIdx.x % n_units;
    }
    const uint32_t slice = sg * LO_RUN + wave;
    const uint32_t lo = u * GH_UNIT + lane;
    const uint32_t lanes_local = n_units * GH_UNIT;
    const O48Lane ol = o48_lane((units ? units[u] : u) * GH_UNIT + lane, out_size / 2);
    if (!__any(lo_lane_wanted(ol, out_size, out_begin, out_count))) return;     // (the same in all waves of the workgroup)

    Texel c = octet_member(2, ol.j, ol.i, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));   // cube.cpp:97

    // FP32 sums of a chunk, pair P: k = 0 (sum, parity sum) of channel P for P < 3; row i of the 2 x 2 product of
    // channel ch at 3 + 2 ch + i; (sum, difference) of the 3 x 3 products at 9 + 9 ch + 3 a + b.
    lo_f32x2 acc[LO_SUMS / 2];
    double dtot[MULTI ? LO_OWN : 1];
#pragma unroll
    for (int k = 0; k < LO_SUMS / 2; k++) acc[k] = lo_f32x2{0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < (MULTI ? LO_OWN : 1); k++) dtot[k] = 0.0;

    const uint32_t in0 = min(n_rec_in, slice * slice_len);
    const uint32_t in1 = min(n_rec_in, in0 + slice_len);
    const uint32_t n_slice_rec = in1 - in0;                          // (0 for the waves past the last slice)

    const lo_f32x2 nx2 = { n.x, n.x }, ny2 = { n.y, n.y }, nz2 = { n.z, n.z }, nxy = { n.x, n.y };

    // The weights of a record from its weight vector g (scalar registers: g.x g.y g.z g.z).
    auto weights = [&](const lo_u32x4& g, LoWeights& W) __attribute__((always_inline)) {
        // (plain vector arithmetic: the compiler then knows the values for canonical and does not quiet them again
        // before the max / min / median below)
        const lo_f32x2 g01 = { __uint_as_float(g[0]), __uint_as_float(g[1]) }, g22 = { __uint_as_float(g[2]), __uint_as_float(g[3]) };
        const lo_f32x2 r0 = nx2 * g01, r1 = ny2 * g01, r2 = nz2 * g01, qq = nxy * g22;   // (n.x g.x, n.x g.y), (n.y g.x, n.y g.y), (n.z g.x, n.z g.y), (n.x g.z, n.y g.z)
        const float m22 = n.z * __uint_as_float(g[2]);
        // a + b and a - b for the permutations in pairs: (P0, P1): a = (r0.x, r0.y), b = (r1.y, r1.x), c = m22;
        // (P2, P3): a = qq.x, b = (r1.y, r1.x), c = (r2.x, r2.y); (P5, P4): a = (r0.y, r0.x), b = qq.y, c = (r2.x, r2.y)
        lo_f32x2 s1[3], s2[3];
        const lo_f32x2 r1s = __builtin_shufflevector(r1, r1, 1, 0), r0s = __builtin_shufflevector(r0, r0, 1, 0);
        const lo_f32x2 qx = __builtin_shufflevector(qq, qq, 0, 0), qy = __builtin_shufflevector(qq, qq, 1, 1);
        s1[0] = r0 + r1s;  s2[0] = r0 - r1s;
        s1[1] = qx + r1s;  s2[1] = qx - r1s;
        s1[2] = r0s + qy;  s2[2] = r0s - qy;
        // |x+y| + |x-y| = 2 max(|x|, |y|), |x+y| - |x-y| = 2 sign(xy) min(|x|, |y|) (see lambertian_split.hip):
        // quartered coefficients w[k] of every permutation from a + b, a - b and c
        lo_f32x2 w0[3], w3[3], w5[3], w6[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const lo_f32x2 cc = t == 0 ? lo_f32x2{m22, m22} : r2;
            const lo_f32x2 p = { __builtin_fmaxf(s1[t].x, cc.x), __builtin_fmaxf(s1[t].y, cc.y) };
            const lo_f32x2 q = { __builtin_fminf(s1[t].x, cc.x), __builtin_fminf(s1[t].y, cc.y) };
            const lo_f32x2 tt = { __builtin_fmaxf(cc.x, fabsf(s2[t].x)), __builtin_fmaxf(cc.y, fabsf(s2[t].y)) };
            const lo_f32x2 md = { __builtin_amdgcn_fmed3f(s2[t].x, -cc.x, cc.x), __builtin_amdgcn_fmed3f(s2[t].y, -cc.y, cc.y) };
            w0[t] = p + tt; w3[t] = p - tt; w5[t] = q + md; w6[t] = q - md;
        }
        // halves: P0 = [0].x, P1 = [0].y, P2 = [1].x, P3 = [1].y, P5 = [2].x, P4 = [2].y
        // characters xy (w3), xz (w5), yz (w6): the even and the odd permutation that carry b to c
        W.wa[0][0] = lo_addsub<0, 1>(w3[0], w3[0]);        // P0, P1
        W.wa[1][0] = lo_addsub<1, 1>(w3[1], w3[2]);        // P3, P4
        W.wa[2][0] = lo_addsub<0, 0>(w3[2], w3[1]);        // P5, P2
        W.wa[0][1] = lo_addsub<0, 1>(w5[2], w5[2]);        // P5, P4
        W.wa[1][1] = lo_addsub<0, 0>(w5[0], w5[1]);        // P0, P2
        W.wa[2][1] = lo_addsub<1, 1>(w5[1], w5[0]);        // P3, P1
        W.wa[0][2] = lo_addsub<1, 0>(w6[1], w6[1]);        // P3, P2
        W.wa[1][2] = lo_addsub<0, 1>(w6[2], w6[0]);        // P5, P1
        W.wa[2][2] = lo_addsub<0, 1>(w6[0], w6[2]);        // P0, P4
        // k = 0 over the permutations
        const float f0 = w0[0].x, f1 = w0[0].y, f2 = w0[1].x, f3 = w0[1].y, f5 = w0[2].x, f4 = w0[2].y;
        const float se = (f0 + f3) + f5, so = (f1 + f2) + f4;
        W.ts = lo_f32x2{ se + so, se - so };
        const float d01 = f0 - f1, a01 = f0 + f1, d45 = f4 - f5, d32 = f3 - f2, a23 = f2 + f3;
        W.f2[0] = lo_f32x2{ d01 + d45, (f1 + f5) - a23 };
        W.f2[1] = lo_f32x2{ d32 + d45, a01 - (f3 + f4) };
    };

#if LO_PREFETCH
    // Every 64 records the wave touches the 64 records after those, a lane each, with a load nobody waits for:
    // a load into LDS, which has no destination register that could be handed to another value while the data is
    // on its way (the words land in s_touch and are never looked at).
    __shared__ float s_touch[64];
    typedef __attribute__((address_space(1))) const void* lo_gptr;
    typedef __attribute__((address_space(3))) void* lo_lptr;
#endif

    LoRec r;
    lo_u32x4 ga, gb;
    LoWeights W;
    for (uint32_t ck = 0; ck < (MULTI ? n_chunks : 1u); ck++) {
        const uint32_t c0 = min(n_slice_rec, ck * chunk_len);
        const uint32_t n_here = min(n_slice_rec - c0, chunk_len);
        if (n_here) {
            const float* __restrict__ rp = rec + (size_t)__builtin_amdgcn_readfirstlane(in0 + c0) * LO_RECF;          // wave-uniform
            const float4* __restrict__ gp = recg + __builtin_amdgcn_readfirstlane(in0 + c0);
            const uint32_t last = n_here - 1u;
            lo_g_load(ga, gp);
            lo_block_load<0>(r, rp); lo_block_load<1>(r, rp); lo_block_load<2>(r, rp); lo_block_load<3>(r, rp); lo_block_load<4>(r, rp);
            lo_g_load(gb, gp + __builtin_amdgcn_readfirstlane(min(1u, last)));
            lo_wait(r, ga, gb);
            weights(ga, W);
            for (uint32_t e = 0; e < n_here; e += 2) {
#if LO_PREFETCH
                if ((e & 63u) == 0u) {
                    const float* pp = rp + (size_t)min(e + 64u + lane, last) * LO_RECF;
                    __builtin_amdgcn_global_load_lds((lo_gptr)pp, (lo_lptr)s_touch, 4, 0, 0);
                }
#endif
                lo_wait(r, ga, gb);                                       // record e, g of e + 1
                lo_fma_record(acc, r, W, rp + (size_t)min(e + 1u, last) * LO_RECF);
                lo_g_load(ga, gp + __builtin_amdgcn_readfirstlane(min(e + 2u, last)));
                weights(gb, W);
                if (e + 1u < n_here) {
                    lo_wait(r, ga, gb);                                   // record e + 1, g of e + 2
                    lo_fma_record(acc, r, W, rp + (size_t)min(e + 2u, last) * LO_RECF);
                    lo_g_load(gb, gp + __builtin_amdgcn_readfirstlane(min(e + 3u, last)));
                    weights(ga, W);
                }
            }
            lo_wait(r, ga, gb);       // nothing is in flight when the chunk ends
        }
        // The four waves' chunk sums meet in LDS, half of the 72 at a time; every wave adds 9 of each half, the
        // waves in their order, to its FP64 totals.
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k = 0; k < LO_SUMS / 4; k++) {
                s_c[wave][2 * k][lane] = acc[(LO_SUMS / 4) * h + k].x;
                s_c[wave][2 * k + 1][lane] = acc[(LO_SUMS / 4) * h + k].y;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < LO_OWN / 2; k++) {
                double v = MULTI ? dtot[MULTI ? (LO_OWN / 2) * h + k : 0] : 0.0;
#pragma unroll
                for (int x = 0; x < LO_RUN; x++) v += (double)s_c[x][(LO_OWN / 2) * wave + k][lane];
                if (MULTI) {
                    dtot[MULTI ? (LO_OWN / 2) * h + k : 0] = v;
                } else if (ol.valid) {          // the only chunk: the group's sum
                    const uint32_t q = (uint32_t)(LO_SUMS / 2) * h + (uint32_t)(LO_OWN / 2) * wave + k;
                    partial[((size_t)sg * LO_SUMS + q) * lanes_local + lo] = (float)v;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < LO_SUMS / 2; k++) acc[k] = lo_f32x2{0.0f, 0.0f};
    }
#if LO_PREFETCH
    // (nothing may still be on its way into the workgroup's LDS when the wave ends)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { const float seen = s_touch[lane]; asm volatile("" : : "v"(seen)); }
#endif

    if (MULTI && ol.valid) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k = 0; k < LO_OWN / 2; k++) {
                const uint32_t q = (uint32_t)(LO_SUMS / 2) * h + (uint32_t)(LO_OWN / 2) * wave + k;
                partial[((size_t)sg * LO_SUMS + q) * lanes_local + lo] = (float)dtot[MULTI ? (LO_OWN / 2) * h + k : 0];
            }
    }
    if (COUNT && lane == 0u) {
        atomicAdd(&counts[3], (unsigned long long)n_slice_rec);
        atomicAdd(&counts[4], (unsigned long long)n_slice_rec * 24ull);      // 24 coefficients per record
    }
}

// The groups of an output unit added up in FP64, 8 of the 72 sums per workgroup (8 interleaved parts, then
// those): grid = units x 9, so that a cube of 33 units still spreads over the chip -- and one workgroup more, which
// adds the first moment's parts in a fixed order (a launch of its own costs 6 us of a 1.5 ms call).
constexpr int LO_REDUCE_PARTS = 8;
constexpr int LO_REDUCE_BATCH = 8;
__global__ __launch_bounds__(64 * LO_REDUCE_PARTS) void lo_reduce(
    const float* __restrict__ groups, uint32_t n_groups, uint32_t out_size, const uint32_t* __restrict__ units, uint32_t n_units,
    uint32_t out_begin, uint32_t out_count, double* __restrict__ esum,
    const double* __restrict__ moment_parts, uint32_t n_parts, double* __restrict__ moment) {
    __shared__ double s_e[LO_REDUCE_PARTS - 1][LO_REDUCE_BATCH][64];
    if (blockIdx.x == n_units * (LO_SUMS / LO_REDUCE_BATCH)) {
        // the first moment: 9 sums over the prep kernel's workgroups, strided over 256 threads, then a fixed tree
        double (*s_m)[256] = (double (*)[256])s_e;
        const uint32_t tid = threadIdx.x;
        if (tid < 256u) {
            double m[9];
#pragma unroll
            for (int k = 0; k < 9; k++) m[k] = 0.0;
            for (uint32_t p = tid; p < n_parts; p += 256u) {
#pragma unroll
                for (int k = 0; k < 9; k++) m[k] += moment_parts[(size_t)p * 9 + k];
            }
#pragma unroll
            for (int k = 0; k < 9; k++) s_m[k][tid] = m[k];
        }
        __syncthreads();
        for (uint32_t step = 128; step > 0; step >>= 1) {
            if (tid < step) {
#pragma unroll
                for (int k = 0; k < 9; k++) s_m[k][tid] += s_m[k][tid + step];
            }
            __syncthreads();
        }
        if (tid < 9) moment[tid] = s_m[tid][0];
        return;
    }
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t u = blockIdx.x / (LO_SUMS / LO_REDUCE_BATCH), batch = blockIdx.x % (LO_SUMS / LO_REDUCE_BATCH);
    const uint32_t lo = u * GH_UNIT + lane;
    const uint32_t lanes_local = n_units * GH_UNIT;
    const O48Lane ol = o48_lane((units ? units[u] : u) * GH_UNIT + lane, out_size / 2);
    if (!__any(lo_lane_wanted(ol, out_size, out_begin, out_count))) return;     // lo_pairs stored nothing for this unit
    double eb[LO_REDUCE_BATCH];
#pragma unroll
    for (int k = 0; k < LO_REDUCE_BATCH; k++) eb[k] = 0.0;
    if (ol.valid)
        for (uint32_t b = part; b < n_groups; b += LO_REDUCE_PARTS) {
#pragma unroll
            for (int k = 0; k < LO_REDUCE_BATCH; k++)
                eb[k] += (double)groups[((size_t)b * LO_SUMS + batch * LO_REDUCE_BATCH + k) * lanes_local + lo];
        }
    if (part > 0) {
#pragma unroll
        for (int k = 0; k < LO_REDUCE_BATCH; k++) s_e[part - 1][k][lane] = eb[k];
    }
    __syncthreads();
    if (part > 0 || !ol.valid) return;
    for (int p = 0; p < LO_REDUCE_PARTS - 1; p++)
#pragma unroll
        for (int k = 0; k < LO_REDUCE_BATCH; k++) eb[k] += s_e[p][k][lane];
#pragma unroll
    for (int k = 0; k < LO_REDUCE_BATCH; k++) esum[(size_t)(batch * LO_REDUCE_BATCH + k) * lanes_local + lo] = eb[k];
}

// The inverse transforms, the odd half, the scale; the member texels that lie in the requested range are stored.
// A workgroup is an output unit; wave p works on the members of permutation p.  flags are those of
// lambertian_octets.hip's kernels over the WHOLE cube's octets: flags[octet] where a texel of the octet is the
// difference of much larger terms, flags[oct_all + block] for its 256-octet block, flags[oct_all + blocks] for any.
__global__ __launch_bounds__(64 * 6) void lo_final(
    const double* __restrict__ esum, const double* __restrict__ moment, uint32_t out_size,
    const uint32_t* __restrict__ units, uint32_t n_units, uint32_t out_begin, uint32_t out_count, double scale, double limit,
    uint32_t oct_blocks, uint32_t tw, uint32_t* __restrict__ flags, unsigned long long* __restrict__ counts, float4* __restrict__ out) {
    LO_TABLES
    const uint32_t lane = threadIdx.x & 63u;
    const int p = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t u = blockIdx.x;
    const uint32_t lo = u * GH_UNIT + lane;
    const uint32_t lanes_local = n_units * GH_UNIT;
    const O48Lane ol = o48_lane((units ? units[u] : u) * GH_UNIT + lane, out_size / 2);
    if (!__any(lo_lane_wanted(ol, out_size, out_begin, out_count))) return;
    if (!ol.valid || (ol.diagonal && (p & 1))) return;
    Texel c = octet_member(2, ol.j, ol.i, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));
    const double n0[3] = { (double)n.x, (double)n.y, (double)n.z };
    const uint32_t hw = out_size / 2;          // tw: the tile edge of the octet numbering
    auto E = [&](int q) { return esum[(size_t)q * lanes_local + lo]; };
    // this permutation's coefficients per channel: k = 0 and the three others, and the size of their terms
    double e0[3], ek[3][3], mag[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const double et = E(2 * ch), es = E(2 * ch + 1);
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int l = 0; l < 2; l++) tr += (double)R2I[p][i][l] * E(2 * (3 + 2 * ch + l) + i);
        // (lo_pairs sums a quarter of the coefficients)
        e0[ch] = (4.0 / 6.0) * (et + (PAR[p] ? -es : es) + 2.0 * tr);
        mag[ch] = (4.0 / 6.0) * (fabs(et) + fabs(es) + 2.0 * fabs(tr));
#pragma unroll
        for (int a = 0; a < 3; a++) {
            int b = 0;
#pragma unroll
            for (int bb = 0; bb < 3; bb++)
                if (EVEN[a][bb] == p || ODD[a][bb] == p) b = bb;
            const double sp = E(2 * (9 + 9 * ch + 3 * a + b)), sm = E(2 * (9 + 9 * ch + 3 * a + b) + 1);
            ek[a][ch] = 2.0 * (PAR[p] ? sp - sm : sp + sm);
            mag[ch] += 2.0 * (fabs(sp) + fabs(sm));
        }
    }
    const uint32_t axis = p < 2 ? 2u : (p < 4 ? 0u : 1u);
    bool bad = false;
#pragma unroll 1
    for (uint32_t s = 0; s < 8; s++) {
        double val[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double even = 0.125 * (e0[ch] + lo_sign(3u, s) * ek[0][ch] + lo_sign(5u, s) * ek[1][ch] + lo_sign(6u, s) * ek[2][ch]);
            const double odd = lo_sign(1u, s) * n0[PERM[p][0]] * moment[ch] + lo_sign(2u, s) * n0[PERM[p][1]] * moment[3 + ch] +
                               lo_sign(4u, s) * n0[PERM[p][2]] * moment[6 + ch];
            const double v = 0.5 * (even + odd);         // the sum the reference forms
            bad = bad || !(0.5 * (0.125 * mag[ch] + fabs(odd)) <= limit * v);
            val[ch] = v * scale;
        }
        Texel t = o48_member(ol.j, ol.i, (uint32_t)p * 8u + s, out_size);
        const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
        if (o - out_begin < out_count) out[o] = make_float4((float)val[0], (float)val[1], (float)val[2], 0.0f);
    }
    if (bad) {
        const uint32_t oct_all = oct_blocks * LO_OCT_BLOCK;
        const uint32_t oo = (p & 1) ? octet_encode(axis, ol.i, ol.j, hw, tw) : octet_encode(axis, ol.j, ol.i, hw, tw);
        flags[oo] = 1u;
        flags[oct_all + oo / LO_OCT_BLOCK] = 1u;
        flags[oct_all + oct_blocks] = 1u;
        if (counts) atomicAdd(&counts[5], 1ull);
    }
}

// ---- lists of units ---------------------------------------------------------------------------------------

uint32_t lambertian_o48_units_total(uint32_t out_size) { return (o48_orbits_total(out_size) + GH_UNIT - 1) / GH_UNIT; }

// Does output unit u hold a texel whose (face, y, x) index lies in [begin, begin + count)?
__host__ __device__ inline bool o48_unit_in_range(uint32_t u, uint32_t out_w, uint32_t begin, uint32_t count) {
    for (uint32_t lane = 0; lane < GH_UNIT; lane++) {
        const O48Lane ol = o48_lane(u * GH_UNIT + lane, out_w / 2);
        if (!ol.valid) continue;
        for (uint32_t m = 0; m < 48u; m++) {
            Texel t = o48_member(ol.j, ol.i, m, out_w);
            if ((t.face * out_w + t.y) * out_w + t.x - begin < count) return true;
        }
    }
    return false;
}

uint32_t lambertian_o48_units_in_range_count(uint32_t out_w, uint32_t begin, uint32_t count) {
    const uint32_t n_units = lambertian_o48_units_total(out_w);
    uint32_t n = 0;
    for (uint32_t u = 0; u < n_units; u++) n += o48_unit_in_range(u, out_w, begin, count) ? 1u : 0u;
    return n;
}

// The list of those units in ascending order (one workgroup; the host knows its length from the count above).
__global__ __launch_bounds__(1024) void o48_units_in_range(uint32_t out_w, uint32_t n_units, uint32_t begin, uint32_t count,
                                                            uint32_t* __restrict__ list) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t u0 = 0; u0 < n_units; u0 += 1024) {
        const uint32_t u = u0 + tid;
        const bool hit = u < n_units && o48_unit_in_range(u, out_w, begin, count);
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

hipError_t lambertian_o48_units_in_range_launch(uint32_t out_w, uint32_t begin, uint32_t count, uint32_t* d_list, hipStream_t stream) {
    o48_units_in_range<<<1, 1024, 0, stream>>>(out_w, lambertian_o48_units_total(out_w), begin, count, d_list);
    return hipGetLastError();
}

// ---- host side -------------------------------------------------------------------------------------

bool lambertian_o48_eligible(uint32_t in_w, uint32_t out_size) { return lambertian_hex_eligible(in_w, out_size); }

// Slices of the input orbits: their number follows from the two cube sizes only (it fixes the order of the
// sums, which must not depend on the part of the cube a call covers).
LambertianO48Plan lambertian_o48_plan(uint32_t in_w, uint32_t out_size, uint32_t n_units) {
    LambertianO48Plan op;
    op.oct = lambertian_oct_plan(in_w, lambertian_oct_blocks_total(out_size));
    op.oct.run = op.oct.n_slices > 1;        // the second pass in its run form (see lambertian_hex_plan)
    op.n_rec = o48_tiled_total(in_w);
    // Up to 1024 slices, fewer where the cube's output units alone make tens of thousands of waves: every group of 4
    // slices writes 72 FP32 sums per output orbit.  1024 -> 128 on one GPU: 512 slices 1.452 ms a call, 640 to 1024
    // 1.436 to 1.447, 1280 1.473, 2048 1.507; an eighth of that cube (5 of its 33 units: a single round of waves)
    // 0.351 ms with 640, 0.329 with 896, 0.302 with 1024, 0.324 with 1280 (tools/shard_times.py).
    const uint32_t units_cube = lambertian_o48_units_total(out_size);
    uint32_t want = 65536u / units_cube;
    want = want < 8u ? 8u : (want > 1024u ? 1024u : want);
    if (want > op.n_rec / 64u) want = op.n_rec / 64u ? op.n_rec / 64u : 1u;     // (small inputs: 64 records a slice at least)
    if (tuning().lamb_oct_slices) want = tuning().lamb_oct_slices;          // tuning knob
    op.slice_len = (op.n_rec + want - 1) / want;
    if (op.slice_len < 2) op.slice_len = 2;
    op.n_slices = (op.n_rec + op.slice_len - 1) / op.slice_len;
    op.n_groups = (op.n_slices + LO_RUN - 1) / LO_RUN;
    // FP32 sums over at most 320 records, and over so few that the input makes some 256 of them: their rounding
    // errors then average out (32 x 32 faces in two chunks of 68: 1.5e-6; in chunks of 8: 3e-7).
    uint32_t chunk = op.n_rec / 256u;
    chunk = chunk < 8u ? 8u : (chunk > LO_CHUNK_MAX ? LO_CHUNK_MAX : chunk);
    op.n_chunks = (op.slice_len + chunk - 1) / chunk;
    op.chunk_len = (op.slice_len + op.n_chunks - 1) / op.n_chunks;
    op.n_units = n_units;
    op.recg_offset = ((size_t)op.n_rec * LO_RECF * sizeof(float) + 255) / 256 * 256;
    const size_t own_rec = op.recg_offset + (size_t)op.n_rec * sizeof(float4);
    op.rec_bytes = own_rec > op.oct.rec_bytes ? own_rec : op.oct.rec_bytes;
    const size_t sums = (size_t)LO_SUMS * n_units * GH_UNIT;
    const size_t fast = sums * (sizeof(float) * op.n_groups + sizeof(double));        // the groups' sums (FP32), then their sum
    const size_t exact = (size_t)24 * op.oct.out_blocks * LO_OCT_BLOCK
                         * (sizeof(double) * op.oct.n_runs + (op.oct.run ? 0 : sizeof(float) * op.oct.n_slices));
    op.work_bytes = lambertian_split_header_bytes(op.oct.n_oct_in, op.oct.out_blocks) + (fast > exact ? fast : exact);
    return op;
}

// The fast pass over the call's output units, then lambertian_octets.hip's kernels over the octets it
// marked (all of the cube's octet blocks are offered; the unmarked ones leave at once).
// d_work: the header of lambertian_split.hip (flags | moment | moment parts), then the sums.
hipError_t lambertian_o48_launch(const LambertianO48Plan& op, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 const uint32_t* d_units, uint32_t out_begin, uint32_t out_count, float* d_out,
                                 void* d_rec, void* d_work, unsigned long long* d_counts, hipStream_t stream,
                                 hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3) {
    const LambertianOctPlan& oc = op.oct;
    const uint32_t prep_blocks = (op.n_rec + 63) / 64;
    const size_t flags_bytes = ((size_t)(oc.out_blocks * LO_OCT_BLOCK + oc.out_blocks + 1) * sizeof(uint32_t) + 255) / 256 * 256;
    uint32_t* d_flags = (uint32_t*)d_work;
    double* d_moment = (double*)((char*)d_work + flags_bytes);
    double* d_parts = (double*)((char*)d_work + flags_bytes + 256);
    char* d_sums = (char*)d_work + lambertian_split_header_bytes(oc.n_oct_in, oc.out_blocks);
    float* d_groups = (float*)d_sums;
    float* d_recl = (float*)d_rec;
    float4* d_recg = (float4*)((char*)d_rec + op.recg_offset);

    lo_prep<<<prep_blocks, LO_PREP_THREADS, 0, stream>>>((const float4*)d_env, in_w, op.n_rec, d_recl, d_recg, d_parts, d_flags,
                                                        oc.out_blocks * LO_OCT_BLOCK + oc.out_blocks + 1);
    const uint32_t affine = (op.n_groups >= 8 && tuning().lamb_oct_affine != 0) ? 1u : 0u;
    const dim3 grid(op.n_units * (affine ? (op.n_groups + 7) / 8 * 8 : op.n_groups));
    if (ev0) (void)hipEventRecord(ev0, stream);
    auto launch = [&](auto kernel) {
        kernel<<<grid, LO_THREADS, 0, stream>>>(d_recl, d_recg, op.n_rec, op.slice_len, op.n_slices, op.chunk_len, op.n_chunks,
            out_size, d_units, op.n_units, affine, out_begin, out_count, d_groups, d_counts);
    };
    if (op.n_chunks > 1) { if (d_counts) launch(lo_pairs<true, true>); else launch(lo_pairs<false, true>); }
    else { if (d_counts) launch(lo_pairs<true, false>); else launch(lo_pairs<false, false>); }
    if (ev1) (void)hipEventRecord(ev1, stream);
    const double scale = 4.0 * 0.31830988618379067154 / ((double)in_w * (double)in_w);      // cube.cpp:126
    // The limit: lo_final's measure of a texel's cancellation takes the sizes of the ORBIT's transformed sums, which
    // is about twice what lambertian_split.hip measures on the octet's own two halves, and the sums themselves are
    // more accurate (a 64 x 64 environment with a light of 1.5e3 ... 3e4 on one face, 32 x 32 out: at twice the
    // limit 0, 0, 160, 654, 768 octets redone and 1.2e-6 at worst, where the hexadecet form at the limit itself
    // redoes 0, 322, 608, 703, 758 and is 1.7e-6 off; at four times the limit this form is 4.2e-6 off).
    double* d_esum = (double*)(d_groups + (size_t)op.n_groups * LO_SUMS * op.n_units * GH_UNIT);
    lo_reduce<<<op.n_units * (LO_SUMS / LO_REDUCE_BATCH) + 1, 64 * LO_REDUCE_PARTS, 0, stream>>>(d_groups, op.n_groups, out_size, d_units, op.n_units,
                                                                                                out_begin, out_count, d_esum, d_parts, prep_blocks, d_moment);
    lo_final<<<op.n_units, 64 * 6, 0, stream>>>(d_esum, d_moment, out_size, d_units, op.n_units,
                                                            out_begin, out_count, scale, 2.0 * (double)tuning().lamb_split_limit,
                                                            oc.out_blocks, lambertian_oct_tile_edge(out_size), d_flags, d_counts, (float4*)d_out);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    // the marked octets with the reference's own weights (the second pass does not count)
    const uint32_t n_oct = 3 * (out_size / 2) * (out_size / 2);
    return lambertian_oct_exact_pass(oc, d_env, in_w, out_size, nullptr, 0, n_oct, out_begin, out_count, d_out, d_rec,
                                     (double*)d_sums, d_flags, nullptr, stream, ev2, ev3);
}

}  // namespace ibl
