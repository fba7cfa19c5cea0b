// Lambertian irradiance through the cube's 16-fold symmetry, with the clamped cosine split into
// its even and odd halves (gfx950).
//
// lambertian_split.hip evaluates integrateLambertian (src/utilities/cube.cpp:95-129) over mirror
// octets: the 8 sign flips leave the reference's cosine unchanged to the bit, the sum over an
// (output octet, input octet) pair is a convolution over the sign group, and its Walsh-Hadamard
// transform leaves four products per channel.  The exchange of x and y leaves every dot product of
// the reference unchanged to the bit as well -- ((0 + a.x b.x) + a.y b.y) + a.z b.z, linear.hpp:100-107:
// a float sum of two products does not depend on their order -- so an octet and its image under the
// exchange (a hexadecet, ibl_device.h) can share more.  With n0, u0 the canonical (all-positive)
// directions of an output and an input hexadecet, R the exchange, D_t the sign flips:
//
//   outputs  O0 = { D_t n0 },  O1 = { D_t R n0 };    inputs  I0 = { D_s u0 },  I1 = { D_s R u0 }
//   (D_t n0) . (D_s u0)     = v00(ts),  v00(r) = rx a  + ry b  + rz c,   a  = nx ux, b  = ny uy, c = nz uz
//   (D_t n0) . (D_s R u0)   = v01(ts),  v01(r) = rx a' + ry b' + rz c,   a' = nx uy, b' = ny ux
//   (D_t R n0) . (D_s u0)   = v01(swap ts),      (D_t R n0) . (D_s R u0) = v00(swap ts)
//
// so the four octet pairs of a hexadecet pair need two sets of weights, not four, and in the
// transformed domain (w^, L^: transforms over the sign group of |v| and of the members' radiance, even
// coefficients k = 0, 3 (xy), 5 (xz), 6 (yz); swap exchanges 5 and 6)
//
//   E0[k] = w00[k] L0[k] + w01[k] L1[k]          E1[k] = w01[swap k] L0[k] + w00[swap k] L1[k]
//
// For k = 0, 3:  E0 + E1 = (w00 + w01)(L0 + L1),  E0 - E1 = (w00 - w01)(L0 - L1): two products; for
// k = 5, 6 eight.  Per hexadecet pair -- 256 texel pairs -- that is 36 multiply-adds per channel set
// (18 packed) and 19 instructions for the coefficients (the two weight sets ride in the two halves of
// packed instructions), where lambertian_split.hip spends 4 x 14.  The linear (odd) half of the clamped
// cosine, the marks for texels that cancel too far and their second pass are lambertian_split.hip's.
//
// Kernel plan
//   lh_prep     one thread per input hexadecet (numbered as units of 64 like the outputs of
//               ggx_hex.hip; lanes above the diagonal of a diagonal tile get a zero record): the
//               128-byte record, and the hexadecet's share of the first moment.
//   lh_moment   the workgroups' shares in a fixed order.
//   lh_pairs    grid = (output units) x (groups of 4 input slices); a workgroup is one unit of 64 output
//               hexadecets and one group, a wave per slice; every wave walks its slice's records with scalar
//               loads, two register sets.  FP32 sums over 128 records, FP64 over the slice; the slices' sums,
//               rounded to FP32, meet in LDS and leave as the group's FP64 sum (no pass over slice sums: as
//               a separate kernel that read 315 MB and took 75 us of a 2.78 ms call).
//   lh_final    all groups in FP64; inverse transforms, the odd half, the scale, the cancellation measured
//               per texel; marks are set per OCTET in the numbering of lambertian_octets.hip, whose kernels
//               redo the marked octets.
#include "ibl_device.h"
#include "ibl_kernels.h"

namespace ibl {

constexpr int LH_RUN = 4;             // slices per workgroup: their sums are added in the workgroup
constexpr int LH_THREADS = 64 * LH_RUN;
constexpr int LH_REC = 8;             // float4 per input hexadecet record
constexpr int LH_SUMS = 24;
#ifndef LH_CHUNK
#define LH_CHUNK 128                  // records per FP32 sum (a power of two)
#endif
constexpr uint32_t LH_OCT_BLOCK = 256;      // the octet blocks of lambertian_octets.hip (marks)

__device__ __forceinline__ double lh_sign(uint32_t k, uint32_t s) { return (__popc(k & s) & 1) ? -1.0 : 1.0; }

// Record (32 floats): g.x g.y g.z 0 | k = 0: (L0+L1, L0-L1) x RGB | k = 3: the same | k = 5: (L0, L1) x RGB |
// k = 6: (L0, L1) x RGB | 0 0 0 0, with L0, L1 the transforms of the radiance of the octet and of its partner
// (0 for a hexadecet on the diagonal, which is its own partner).
// (The kernel also clears the call's marks: n_flags words.)
__global__ __launch_bounds__(256) void lh_prep(const float4* __restrict__ env, uint32_t w, uint32_t n_rec,
                                                float4* __restrict__ rec, double* __restrict__ moment_parts,
                                                uint32_t* __restrict__ flags, uint32_t n_flags) {
    // (one buffer for both uses: the records on their way out, then the moment's tree)
    __shared__ float4 s_rec[256 * (LH_REC + 1)];
    double (*s_m)[256] = (double (*)[256])s_rec;
    const uint32_t tid = threadIdx.x;
    const uint32_t k = blockIdx.x * blockDim.x + tid;
    for (uint32_t i = k; i < n_flags; i += gridDim.x * blockDim.x) flags[i] = 0u;
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; q++) m[q] = 0.0;
    float out[32];
#pragma unroll
    for (int q = 0; q < 32; q++) out[q] = 0.0f;
    if (k < n_rec) {
        const HexLane hl = hex_unit_lane(k / GH_UNIT, k % GH_UNIT, w / 2);
        if (hl.valid) {
            Texel c = octet_member(hl.axis, hl.j, hl.i, 0, w, w);
            F3 d = ref_normal_from(c.face, c.x, c.y, w, w);       // all components > 0
            float l2 = ref_dot(d, d);
            float inv = 1.0f / (l2 * l2);                         // cube.cpp:116-117
            const float g[3] = { d.x * inv, d.y * inv, d.z * inv };
            const float gp[3] = { g[1], g[0], g[2] };             // the partner's: the same numbers, x and y exchanged
            out[0] = g[0]; out[1] = g[1]; out[2] = g[2];
            double L[2][8][3];
#pragma unroll
            for (uint32_t s = 0; s < 16; s++) {
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (s < 8u || !hl.diagonal) {
                    Texel t = hex_member(hl.axis, hl.j, hl.i, s, w);
                    v = env[(t.face * w + t.y) * w + t.x];        // cube.cpp:104 row stride (square faces)
                }
                L[s >> 3][s & 7u][0] = v.x; L[s >> 3][s & 7u][1] = v.y; L[s >> 3][s & 7u][2] = v.z;
            }
            const uint32_t ks[4] = { 0u, 3u, 5u, 6u };
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    double v0 = 0.0, v1 = 0.0;
#pragma unroll
                    for (uint32_t s = 0; s < 8; s++) { v0 += lh_sign(ks[q], s) * L[0][s][ch]; v1 += lh_sign(ks[q], s) * L[1][s][ch]; }
                    if (q < 2) { out[4 + 6 * q + 2 * ch] = (float)(v0 + v1); out[4 + 6 * q + 2 * ch + 1] = (float)(v0 - v1); }
                    else { out[4 + 6 * q + 2 * ch] = (float)v0; out[4 + 6 * q + 2 * ch + 1] = (float)v1; }
                }
            // first moment: member s of the octet points to (+-gx, +-gy, +-gz), of the partner to (+-gy, +-gx, +-gz)
#pragma unroll
            for (int ax = 0; ax < 3; ax++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    double v0 = 0.0, v1 = 0.0;
#pragma unroll
                    for (uint32_t s = 0; s < 8; s++) { v0 += lh_sign(1u << ax, s) * L[0][s][ch]; v1 += lh_sign(1u << ax, s) * L[1][s][ch]; }
                    m[ax * 3 + ch] = (double)g[ax] * v0 + (double)gp[ax] * v1;
                }
        }
    }
    // The workgroup's 256 records leave through LDS, so that neighbouring threads store neighbouring 16-byte
    // words (a thread storing its own 128 bytes writes eight quarter-filled lines per instruction; rows of
    // 9 words keep the LDS writes off each other's banks).
    {
#pragma unroll
        for (int q = 0; q < LH_REC; q++)
            s_rec[tid * (LH_REC + 1) + q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        __syncthreads();
        const size_t base = (size_t)blockIdx.x * 256 * LH_REC, end = (size_t)n_rec * LH_REC;
#pragma unroll
        for (int q = 0; q < LH_REC; q++) {
            const uint32_t f = (uint32_t)q * 256u + tid;
            if (base + f < end) rec[base + f] = s_rec[(f / LH_REC) * (LH_REC + 1) + f % LH_REC];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 9; q++) s_m[q][tid] = m[q];
    __syncthreads();
    for (uint32_t step = 128; step > 0; step >>= 1) {          // fixed tree
        if (tid < step) {
#pragma unroll
            for (int q = 0; q < 9; q++) s_m[q][tid] += s_m[q][tid + step];
        }
        __syncthreads();
    }
    if (tid < 9) moment_parts[(size_t)blockIdx.x * 9 + tid] = s_m[tid][0];
}

__global__ __launch_bounds__(256) void lh_moment(const double* __restrict__ parts, uint32_t n_parts, double* __restrict__ moment) {
    __shared__ double s_m[9][256];
    const uint32_t tid = threadIdx.x;
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; q++) m[q] = 0.0;
    for (uint32_t p = tid; p < n_parts; p += 256) {
#pragma unroll
        for (int q = 0; q < 9; q++) m[q] += parts[(size_t)p * 9 + q];
    }
#pragma unroll
    for (int q = 0; q < 9; q++) s_m[q][tid] = m[q];
    __syncthreads();
    for (uint32_t step = 128; step > 0; step >>= 1) {
        if (tid < step) {
#pragma unroll
            for (int q = 0; q < 9; q++) s_m[q][tid] += s_m[q][tid + step];
        }
        __syncthreads();
    }
    if (tid < 9) moment[tid] = s_m[tid][0];
}

// A record in scalar registers (load and wait written out, as in lambertian_split.hip).
typedef uint32_t lh_u32x16 __attribute__((ext_vector_type(16)));
struct LhRec { lh_u32x16 a, b; };
__device__ __forceinline__ void lh_rec_load(LhRec& r, const float4* __restrict__ p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r.a), "=&s"(r.b) : "s"(p));
}
__device__ __forceinline__ void lh_rec_wait(LhRec& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b));
}

#ifndef LH_MIN_WAVES
#define LH_MIN_WAVES 1
#endif
#ifndef LH_PREFETCH
#define LH_PREFETCH 1          // the waves touch records ahead of their scalar loads (see lambertian_split.hip); 0: they do not
#endif

// Does the unit hold a texel of the (face, y, x) range?  (Whole cubes and shards skip the test.)
__device__ __forceinline__ bool lh_lane_wanted(const HexLane& hl, uint32_t out_size, uint32_t out_begin, uint32_t out_count) {
    if (!hl.valid) return false;
    if (out_begin == 0u && out_count >= 6u * out_size * out_size) return true;
    bool wanted = false;
    for (uint32_t s = 0; s < (hl.diagonal ? 8u : 16u); s++) {
        Texel t = hex_member(hl.axis, hl.j, hl.i, s, out_size);
        const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
        wanted = wanted || (o - out_begin < out_count);
    }
    return wanted;
}

// COUNT: the wavefronts also count the records they walk.
template <bool COUNT>
__global__ __launch_bounds__(LH_THREADS, LH_MIN_WAVES) void lh_pairs(
    const float4* __restrict__ rec, uint32_t n_rec_in, uint32_t slice_len, uint32_t n_slices,
    uint32_t out_size, const uint32_t* __restrict__ units, uint32_t n_units, uint32_t out_groups, uint32_t xcd_affine,
    uint32_t out_begin, uint32_t out_count, double* __restrict__ partial, unsigned long long* __restrict__ counts) {
    // The workgroup: one output unit, LH_RUN consecutive slices, a wave each; their sums meet in LDS at the end.
    __shared__ float s_slice[LH_RUN - 1][LH_SUMS][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: the record addresses hang on it)
    const uint32_t n_sgroups = (n_slices + LH_RUN - 1) / LH_RUN;
    uint32_t sg, u;
    if (xcd_affine) {                                   // XCD k works on the slice groups k, k + 8, ...
        const uint32_t seq = blockIdx.x >> 3;
        sg = (blockIdx.x & 7u) + 8u * (seq / out_groups);
        u = seq % out_groups;
        if (sg >= n_sgroups) return;
    } else {
        sg = blockIdx.x / out_groups;
        u = blockIdx.x % out_groups;
    }
    const uint32_t slice = sg * LH_RUN + wave;
    const uint32_t lo = u * GH_UNIT + lane;
    const uint32_t hex_local = n_units * GH_UNIT;
    const HexLane hl = hex_unit_lane(units ? units[u] : u, lane, out_size / 2);
    if (!__any(lh_lane_wanted(hl, out_size, out_begin, out_count))) return;     // (the same in all waves of the workgroup)

    Texel c = octet_member(hl.axis, hl.j, hl.i, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));   // cube.cpp:97

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // FP32 sums of a chunk: (E0 + E1, E0 - E1) per channel for k = 0 and 3, (E0, E1) per channel for k = 5 and 6
    f32x2 acc[4][3];
    double dacc[LH_SUMS];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) acc[q][ch] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < LH_SUMS; q++) dacc[q] = 0.0;

    const uint32_t in0 = min(n_rec_in, slice * slice_len);
    const uint32_t in1 = min(n_rec_in, in0 + slice_len);
    const uint32_t n_rec = in1 - in0;                                // (0 for the waves past the last slice)
    const float4* __restrict__ rp = rec + (size_t)min(in0, n_rec_in - 1u) * LH_REC;      // wave-uniform

    const f32x2 nx = { n.x, n.x }, ny = { n.y, n.y };
    auto eval = [&](const LhRec& r) __attribute__((always_inline)) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        auto spair = [&](int at) __attribute__((always_inline)) {
            return at < 16 ? u32x2{r.a[at & 15], r.a[(at + 1) & 15]} : u32x2{r.b[(at - 16) & 15], r.b[(at - 15) & 15]};
        };
        // The two weight sets in the halves of packed instructions: (a, a') = nx (gx, gy), (b, b') = ny (gy, gx).
        f32x2 aa, bb;
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(aa) : "v"(nx), "s"(spair(0)));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(bb) : "v"(ny), "s"(spair(0)));
        const float cc = n.z * __uint_as_float(r.a[2]);
        const f32x2 s1 = aa + bb, s2 = aa - bb;
        // |x+y| + |x-y| = 2 max(|x|, |y|), |x+y| - |x-y| = 2 sign(xy) min(|x|, |y|) (see lambertian_split.hip):
        // halved coefficients of both sets from a+b, a-b and c
        const f32x2 p = { __builtin_fmaxf(s1.x, cc), __builtin_fmaxf(s1.y, cc) };
        const f32x2 q = { __builtin_fminf(s1.x, cc), __builtin_fminf(s1.y, cc) };
        const f32x2 t = { __builtin_fmaxf(cc, fabsf(s2.x)), __builtin_fmaxf(cc, fabsf(s2.y)) };
        const f32x2 m = { __builtin_amdgcn_fmed3f(s2.x, -cc, cc), __builtin_amdgcn_fmed3f(s2.y, -cc, cc) };
        const f32x2 w0 = p + t, w3 = p - t;                           // (w00[k], w01[k]) / 2 for k = 0, 3
        // k = 5, 6:  A = w00[5] = q0 + m0, C = w00[6] = q0 - m0, B = w01[5] = q1 + m1, D = w01[6] = q1 - m1
        f32x2 X, Y, W0, W3;                                           // X = (A, D), Y = (C, B); W = (w00 + w01, w00 - w01)
        asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(X) : "v"(q), "v"(m));
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(Y) : "v"(q), "v"(m));
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(W0) : "v"(w0));
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(W3) : "v"(w3));
        // Packed multiply-adds with the record data straight from scalar registers.
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0][ch]) : "v"(W0), "s"(spair(4 + 2 * ch)));
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[1][ch]) : "v"(W3), "s"(spair(10 + 2 * ch)));
            // (E0[5], E1[5]) += (A, D) L0[5] + (B, C) L1[5]
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[2][ch]) : "v"(X), "s"(spair(16 + 2 * ch)));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc[2][ch]) : "v"(Y), "s"(spair(16 + 2 * ch)));
            // (E0[6], E1[6]) += (C, B) L0[6] + (D, A) L1[6]
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[3][ch]) : "v"(Y), "s"(spair(22 + 2 * ch)));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc[3][ch]) : "v"(X), "s"(spair(22 + 2 * ch)));
        }
    };
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dacc[6 * q + ch] += (double)acc[q][ch].x;
                dacc[6 * q + 3 + ch] += (double)acc[q][ch].y;
                acc[q][ch] = f32x2{0.0f, 0.0f};
            }
    };

    LhRec ra, rb;
    const uint32_t last = n_rec ? n_rec - 1u : 0u;
#if LH_PREFETCH
    // Every 64 records the wave touches the 64 records after those, a lane each, with a load nobody waits for:
    // a load into LDS, which has no destination register that could be handed to another value while the data is
    // on its way (the words land in s_touch and are never looked at).
    __shared__ float s_touch[64];
    typedef __attribute__((address_space(1))) const void* lh_gptr;
    typedef __attribute__((address_space(3))) void* lh_lptr;
#endif
    lh_rec_load(ra, rp);
    for (uint32_t e = 0; e < n_rec; e += 2) {
#if LH_PREFETCH
        if ((e & 63u) == 0u) {
            const float4* pp = rp + (size_t)min(e + 64u + lane, last) * LH_REC;
            __builtin_amdgcn_global_load_lds((lh_gptr)pp, (lh_lptr)s_touch, 4, 0, 0);
        }
#endif
        lh_rec_wait(ra);
        lh_rec_load(rb, rp + (size_t)min(e + 1u, last) * LH_REC);
        eval(ra);
        lh_rec_wait(rb);
        lh_rec_load(ra, rp + (size_t)min(e + 2u, last) * LH_REC);
        if (e + 1u < n_rec) eval(rb);
        if ((e & (uint32_t)(LH_CHUNK - 1)) == (uint32_t)(LH_CHUNK - 2)) flush();      // every LH_CHUNK records
    }
    lh_rec_wait(ra);       // nothing is in flight when the wave ends
#if LH_PREFETCH
    // (nothing may still be on its way into the workgroup's LDS when the wave ends)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { const float seen = s_touch[lane]; asm volatile("" : : "v"(seen)); }
#endif
    flush();

    // The slices' sums as FP32, added in slice order in FP64 by the first wave: the group's sum.
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < LH_SUMS; q++) s_slice[wave - 1][q][lane] = (float)dacc[q];
    }
    __syncthreads();
    if (wave == 0 && hl.valid) {
#pragma unroll
        for (int q = 0; q < LH_SUMS; q++) {
            double v = (double)(float)dacc[q];
#pragma unroll
            for (int k = 0; k < LH_RUN - 1; k++) v += (double)s_slice[k][q][lane];
            partial[((size_t)sg * LH_SUMS + q) * hex_local + lo] = v;
        }
    }
    if (COUNT && lane == 0u) {
        atomicAdd(&counts[3], (unsigned long long)n_rec);
        atomicAdd(&counts[4], (unsigned long long)n_rec * 8ull);      // eight coefficients per record
    }
}

// The runs of an output hexadecet added up (8 interleaved parts, then those, as ls_final), the inverse
// transforms, the odd half, the scale; the member texels that lie in the requested range are stored.
// flags are those of lambertian_octets.hip's kernels over the WHOLE cube's octets: flags[octet] where a
// texel of the octet is the difference of much larger halves, flags[oct_all + block] for its 256-octet
// block, flags[oct_all + blocks] for any.
constexpr int LH_FINAL_PARTS = 8;
__global__ __launch_bounds__(64 * LH_FINAL_PARTS) void lh_final(
    const double* __restrict__ runs, uint32_t n_runs, const double* __restrict__ moment, uint32_t out_size,
    const uint32_t* __restrict__ units, uint32_t n_units, uint32_t out_begin, uint32_t out_count, double scale, double limit,
    uint32_t oct_blocks, uint32_t tw, uint32_t* __restrict__ flags, unsigned long long* __restrict__ counts, float4* __restrict__ out) {
    __shared__ double s_e[LH_FINAL_PARTS - 1][LH_SUMS][64];
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t u = blockIdx.x;
    const uint32_t lo = u * GH_UNIT + lane;
    const uint32_t hex_local = n_units * GH_UNIT;
    const HexLane hl = hex_unit_lane(units ? units[u] : u, lane, out_size / 2);
    if (!__any(lh_lane_wanted(hl, out_size, out_begin, out_count))) return;     // lh_pairs stored nothing for this unit
    double e[LH_SUMS];
#pragma unroll
    for (int q = 0; q < LH_SUMS; q++) e[q] = 0.0;
    if (hl.valid)
        for (uint32_t b = part; b < n_runs; b += LH_FINAL_PARTS) {
#pragma unroll
            for (int q = 0; q < LH_SUMS; q++) e[q] += runs[((size_t)b * LH_SUMS + q) * hex_local + lo];
        }
    if (part > 0) {
#pragma unroll
        for (int q = 0; q < LH_SUMS; q++) s_e[part - 1][q][lane] = e[q];
    }
    __syncthreads();
    if (part > 0 || !hl.valid) return;
    for (int p = 0; p < LH_FINAL_PARTS - 1; p++)
#pragma unroll
        for (int q = 0; q < LH_SUMS; q++) e[q] += s_e[p][q][lane];
    Texel c = octet_member(hl.axis, hl.j, hl.i, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));
    const uint32_t hw = out_size / 2;          // tw: the tile edge of the octet numbering
    uint32_t n_bad = 0;
#pragma unroll 1
    for (uint32_t half = 0; half < (hl.diagonal ? 1u : 2u); half++) {
        const double nn[3] = { half ? (double)n.y : (double)n.x, half ? (double)n.x : (double)n.y, (double)n.z };
        bool bad = false;
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) {
            double val[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                // (lh_pairs sums half the coefficients)
                const double e0 = 0.5 * (half ? e[ch] - e[3 + ch] : e[ch] + e[3 + ch]);
                const double e3 = 0.5 * (half ? e[6 + ch] - e[9 + ch] : e[6 + ch] + e[9 + ch]);
                const double e5 = half ? e[15 + ch] : e[12 + ch];
                const double e6 = half ? e[21 + ch] : e[18 + ch];
                const double even = 2.0 * (e0 + lh_sign(3u, s) * e3 + lh_sign(5u, s) * e5 + lh_sign(6u, s) * e6);
                const double odd = 4.0 * (lh_sign(1u, s) * nn[0] * moment[ch] + lh_sign(2u, s) * nn[1] * moment[3 + ch] +
                                          lh_sign(4u, s) * nn[2] * moment[6 + ch]);
                const double v = even + odd;               // 8 x the sum the reference forms
                bad = bad || !(fabs(even) + fabs(odd) <= limit * v);
                val[ch] = v * (scale * 0.125);
            }
            Texel t = hex_member(hl.axis, hl.j, hl.i, s | (half << 3), out_size);
            const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
            if (o - out_begin < out_count) out[o] = make_float4((float)val[0], (float)val[1], (float)val[2], 0.0f);
        }
        if (bad) {
            const uint32_t oct_all = oct_blocks * LH_OCT_BLOCK;
            const uint32_t oo = half ? octet_encode(hl.axis == 0 ? 1u : 2u, hl.i, hl.j, hw, tw) : octet_encode(hl.axis, hl.j, hl.i, hw, tw);
            flags[oo] = 1u;
            flags[oct_all + oo / LH_OCT_BLOCK] = 1u;
            flags[oct_all + oct_blocks] = 1u;
            n_bad++;
        }
    }
    if (n_bad && counts) atomicAdd(&counts[5], (unsigned long long)n_bad);
}

// ---- host side -------------------------------------------------------------------------------------

bool lambertian_hex_eligible(uint32_t in_w, uint32_t out_size) {
    return in_w % 16u == 0u && in_w >= 32u && out_size % 16u == 0u && out_size >= 32u
        && cube_is_mirror_exact(in_w) && cube_is_mirror_exact(out_size);
}

// Slices of the input hexadecets: as many as the octet plan has for this input size (the count fixes
// the order of the sums, so it depends on the input size only).
LambertianHexPlan lambertian_hex_plan(uint32_t in_w, uint32_t out_size, uint32_t n_units) {
    LambertianHexPlan hp;
    hp.oct = lambertian_oct_plan(in_w, lambertian_oct_blocks_total(out_size));
    // The second pass in its run form: it is offered every octet block of the cube whatever the call covers, so
    // an eighth of the sums (and one launch less when nothing is marked, the usual case) is worth more than the
    // slice form's 15 % on the marked blocks; the two forms give the same bits.
    hp.oct.run = hp.oct.n_slices > 1;
    hp.n_rec = hex_units_total(in_w) * GH_UNIT;
    // At most 341 slices, 512 from 1024 x 1024 faces on (or what the knob says): on the 1024 -> 128 cube 512 beat
    // 430 by 2-6 % on 1, 2, 4 and 8 shards alike (a wave is a fifth shorter, and an eighth of that cube is a single
    // round of waves and a tail); 683 to 1024 gain nothing more and write more sums (tools/lamb_slices_sweep.py).
    const uint32_t max_slices = tuning().lamb_oct_slices ? hp.oct.n_slices : (hp.n_rec >= 262144u ? 512u : hp.oct.n_slices);
    hp.slice_len = (hp.n_rec + max_slices - 1) / max_slices;
    if (hp.slice_len < 2) hp.slice_len = 2;
    hp.n_slices = (hp.n_rec + hp.slice_len - 1) / hp.slice_len;
    hp.n_runs = (hp.n_slices + LH_RUN - 1) / LH_RUN;
    hp.n_units = n_units;
    const size_t hex_rec = (size_t)hp.n_rec * LH_REC * sizeof(float4);
    hp.rec_bytes = hex_rec > hp.oct.rec_bytes ? hex_rec : hp.oct.rec_bytes;
    const size_t sums = (size_t)LH_SUMS * n_units * GH_UNIT;
    const size_t fast = sums * sizeof(double) * hp.n_runs;
    const size_t exact = (size_t)24 * hp.oct.out_blocks * LH_OCT_BLOCK
                         * (sizeof(double) * hp.oct.n_runs + (hp.oct.run ? 0 : sizeof(float) * hp.oct.n_slices));
    hp.work_bytes = lambertian_split_header_bytes(hp.oct.n_oct_in, hp.oct.out_blocks) + (fast > exact ? fast : exact);
    return hp;
}

// The fast pass over the call's output units, then lambertian_octets.hip's kernels over the octets it
// marked (all of the cube's octet blocks are offered; the unmarked ones leave at once).
// d_work: the header of lambertian_split.hip (flags | moment | moment parts), then the sums.
hipError_t lambertian_hex_launch(const LambertianHexPlan& hp, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 const uint32_t* d_units, uint32_t out_begin, uint32_t out_count, float* d_out,
                                 void* d_rec, void* d_work, unsigned long long* d_counts, hipStream_t stream,
                                 hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3) {
    const LambertianOctPlan& op = hp.oct;
    const uint32_t prep_blocks = (hp.n_rec + 255) / 256;
    const size_t flags_bytes = ((size_t)(op.out_blocks * LH_OCT_BLOCK + op.out_blocks + 1) * sizeof(uint32_t) + 255) / 256 * 256;
    uint32_t* d_flags = (uint32_t*)d_work;
    double* d_moment = (double*)((char*)d_work + flags_bytes);
    double* d_parts = (double*)((char*)d_work + flags_bytes + 256);
    char* d_sums = (char*)d_work + lambertian_split_header_bytes(op.n_oct_in, op.out_blocks);
    double* d_runs = (double*)d_sums;

    lh_prep<<<prep_blocks, 256, 0, stream>>>((const float4*)d_env, in_w, hp.n_rec, (float4*)d_rec, d_parts, d_flags,
                                             op.out_blocks * LH_OCT_BLOCK + op.out_blocks + 1);
    lh_moment<<<1, 256, 0, stream>>>(d_parts, prep_blocks, d_moment);
    const uint32_t out_groups = hp.n_units;
    const uint32_t affine = (hp.n_runs >= 8 && tuning().lamb_oct_affine != 0) ? 1u : 0u;
    const dim3 grid(out_groups * (affine ? (hp.n_runs + 7) / 8 * 8 : hp.n_runs));
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (d_counts)
        lh_pairs<true><<<grid, LH_THREADS, 0, stream>>>((const float4*)d_rec, hp.n_rec, hp.slice_len, hp.n_slices, out_size,
            d_units, hp.n_units, out_groups, affine, out_begin, out_count, d_runs, d_counts);
    else
        lh_pairs<false><<<grid, LH_THREADS, 0, stream>>>((const float4*)d_rec, hp.n_rec, hp.slice_len, hp.n_slices, out_size,
            d_units, hp.n_units, out_groups, affine, out_begin, out_count, d_runs, d_counts);
    if (ev1) (void)hipEventRecord(ev1, stream);
    const double scale = 4.0 * 0.31830988618379067154 / ((double)in_w * (double)in_w);      // cube.cpp:126
    lh_final<<<hp.n_units, 64 * LH_FINAL_PARTS, 0, stream>>>(d_runs, hp.n_runs, d_moment, out_size, d_units, hp.n_units,
                                                            out_begin, out_count, scale, (double)tuning().lamb_split_limit,
                                                            op.out_blocks, lambertian_oct_tile_edge(out_size), d_flags, d_counts, (float4*)d_out);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    // the marked octets with the reference's own weights (the second pass does not count)
    const uint32_t n_oct = 3 * (out_size / 2) * (out_size / 2);
    return lambertian_oct_exact_pass(op, d_env, in_w, out_size, nullptr, 0, n_oct, out_begin, out_count, d_out, d_rec,
                                     (double*)d_sums, d_flags, nullptr, stream, ev2, ev3);
}

}  // namespace ibl
