// Lambertian irradiance, mirror octets, with the clamped cosine split into its even and odd
// halves (gfx950).
//
// Same sum as lambertian_octets.hip (integrateLambertian, src/utilities/cube.cpp:95-129): per
// (output octet, input octet) the cosine of output member so with input member si is
//     v_t = tx*a + ty*b + tz*c,   a = nx*gx, b = ny*gy, c = nz*gz,   t = so ^ si,
// and the sum the reference forms is  out[so] = sum_t max(0, v_t) * L[so ^ t]  over the 8 sign
// patterns t (cube.cpp:114-120).  Two facts cut the arithmetic by a factor of four:
//
//   max(0, v) = (v + |v|) / 2.   The linear half needs no pairs at all: summed over the input it
//   is  n . M  with the first moment M = sum over input texels of g * L (9 numbers for the whole
//   cube: ls_prep / ls_moment).  Only |v| is left for the pair sum, and |v_t| = |v_(7-t)|.
//
//   out[so] = sum_t w_t L[so ^ t] is a convolution over the sign group, so the Walsh-Hadamard
//   transform over the 8 members turns it into 8 independent products  O^[k] = w^[k] * L^[k].
//   With w_t = |v_t| the odd-parity k vanish and four coefficients remain:
//       w^[0] = |v0| + |v4| + |v1| + |v2|      w^[3] = |v0| + |v4| - |v1| - |v2|
//       w^[5] = |v0| - |v4| - |v1| + |v2|      w^[6] = |v0| - |v4| + |v1| - |v2|
//   (v0 = a+b+c > 0, v4 = a+b-c, v2 = a-b+c, v1 = -a+b+c).  The records hold L^[k] for those four
//   k (ls_prep), so one record costs 12 vector operations for the coefficients and 6 packed
//   multiply-adds per output octet (ls_pairs; issued for two records at a time, 8 of the 12 as
//   4 packed instructions), where lambertian_oct_pairs spends 11 and then 13 more for every sign
//   pattern that is live in the wave (4.2 on average).
//
// What this costs is exactness of the zero: where an output texel sees much less light than its
// mirror images and its antipode, its value is the small difference of the two halves, and their
// FP32 rounding (about 1e-7 of the larger) no longer vanishes with it.  ls_final therefore measures
// the cancellation of every texel it combines, (even + |odd|) / (even + odd), and marks the
// output octets where it exceeds a limit (16: measured error 1.2e-7 per unit of cancellation, so
// at most ~2e-6 of the value); the marked
// octets are then recomputed by lambertian_oct_pairs with the reference's own weights
// (lambertian_oct_launch works on the 256-octet blocks that hold one, and stores the marked
// octets only).  The mark depends on the octet's own 8 texels alone, never on the range or shard
// a call covers, so a texel's value still does not depend on how a bake is split.
//
// Kernel plan
//   ls_prep      one thread per input octet: the record (folded direction g, L^[0,3,5,6] x RGB,
//                64 bytes; two records share a 128-byte block laid out for packed instructions),
//                and the octet's share of the first moment, summed per workgroup in FP64.
//   ls_moment    one workgroup: the workgroups' shares in a fixed order.
//   ls_pairs     grid = (blocks of 256 output octets) x (input slices), as lambertian_oct_pairs;
//                every wave works alone on 64 output octets and walks the slice's records with
//                wave-uniform (scalar) loads, two register sets, the next block of two records in
//                flight while one is evaluated: no LDS, no barrier.  FP32 sums over 128 records
//                (blue: the even and the odd records apart), FP64 over the slice, stored as FP32
//                slice sums.
//   ls_reduce / ls_final   runs of 8 slices, then all runs (8 interleaved parts, then those), in FP64;
//                inverse transform, the odd half added, scaled, cancellation measured, stored where
//                the range asks.
#include "ibl_device.h"
#include "ibl_kernels.h"

namespace ibl {

constexpr int LS_THREADS = 256;       // 4 waves, each its own 64 output octets of the block
constexpr int LS_REC = 4;             // float4 per input octet record
constexpr int LS_SUMS = 12;           // 4 coefficients x RGB
#ifndef LS_CHUNK
#define LS_CHUNK 128                  // records per FP32 sum (a power of two)
#endif
constexpr int LS_RUN = 8;             // slices per run (the same grouping as lambertian_octets.hip)

__host__ __device__ inline uint32_t ls_tile_edge(uint32_t hw) {
    uint32_t tw = hw < 8 ? hw : 8;
    while (hw % tw) tw--;
    return tw;
}

// sign of member s under coefficient k: (-1)^popcount(k & s)
__device__ __forceinline__ double ls_sign(uint32_t k, uint32_t s) { return (__popc(k & s) & 1) ? -1.0 : 1.0; }

__global__ __launch_bounds__(256) void ls_prep(const float4* __restrict__ env, uint32_t w, uint32_t n_oct,
                                                float4* __restrict__ rec, double* __restrict__ moment_parts) {
    __shared__ double s_m[9][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = blockIdx.x * blockDim.x + tid;
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; q++) m[q] = 0.0;
    float g[3] = {0.0f, 0.0f, 0.0f};
    float e[LS_SUMS];
#pragma unroll
    for (int q = 0; q < LS_SUMS; q++) e[q] = 0.0f;        // (past the end: the zero record that pads an odd count)
    if (k < n_oct) {
        const uint32_t hw = w / 2;
        const uint32_t axis = k / (hw * hw);
        const uint32_t rem = k - axis * (hw * hw);
        const uint32_t j = rem / hw, i = rem - j * hw;
        Texel c = octet_member(axis, j, i, 0, w, w);
        F3 d = ref_normal_from(c.face, c.x, c.y, w, w);       // all components > 0
        float l2 = ref_dot(d, d);
        float inv = 1.0f / (l2 * l2);                         // cube.cpp:116-117
        g[0] = d.x * inv; g[1] = d.y * inv; g[2] = d.z * inv;
        double L[8][3];
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) {
            Texel t = octet_member(axis, j, i, s, w, w);
            float4 v = env[(t.face * w + t.y) * w + t.x];     // cube.cpp:104 row stride (square faces)
            L[s][0] = v.x; L[s][1] = v.y; L[s][2] = v.z;
        }
        const uint32_t ks[4] = { 0u, 3u, 5u, 6u };
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                double v = 0.0;
#pragma unroll
                for (uint32_t s = 0; s < 8; s++) v += ls_sign(ks[q], s) * L[s][ch];
                e[q * 3 + ch] = (float)v;
            }
        // first moment: member s points to (+-gx, +-gy, +-gz), minus where its bit is set
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                double v = 0.0;
#pragma unroll
                for (uint32_t s = 0; s < 8; s++) v += ls_sign(1u << ax, s) * L[s][ch];
                m[ax * 3 + ch] = (double)g[ax] * v;
            }
    }
    // Two records to a 128-byte block, laid out for packed instructions on the pair:
    //   0-5    g.x of both, g.y of both, g.z of both        6-7  unused (0)
    //   8+6q   (red, green) of coefficient q of the first record, the same of the second,
    //          blue of both                                  (q = 0..3)
    // The two threads of a block swap what the other one stores and each writes one 64-byte half
    // with four 16-byte stores (as 16 scattered 4-byte stores per thread this kernel took 43 us for a
    // 1024 x 1024 face, most of what an eighth of a cube spends outside ls_pairs).
    {
        const uint32_t h = k & 1u;
        float mine[16], other[16], blk[32];
        mine[0] = g[0]; mine[1] = g[1]; mine[2] = g[2]; mine[3] = 0.0f;
#pragma unroll
        for (int q = 0; q < LS_SUMS; q++) mine[4 + q] = e[q];
#pragma unroll
        for (int q = 0; q < 16; q++) other[q] = __shfl_xor(mine[q], 1);
        const float* r0 = h ? other : mine;        // the block's first and second record
        const float* r1 = h ? mine : other;
        blk[0] = r0[0]; blk[1] = r1[0]; blk[2] = r0[1]; blk[3] = r1[1]; blk[4] = r0[2]; blk[5] = r1[2];
        blk[6] = 0.0f; blk[7] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            blk[8 + 6 * q] = r0[4 + 3 * q]; blk[8 + 6 * q + 1] = r0[5 + 3 * q];
            blk[8 + 6 * q + 2] = r1[4 + 3 * q]; blk[8 + 6 * q + 3] = r1[5 + 3 * q];
            blk[8 + 6 * q + 4] = r0[6 + 3 * q]; blk[8 + 6 * q + 5] = r1[6 + 3 * q];
        }
        if ((k & ~1u) < n_oct) {
            float4* dst = rec + (size_t)(k >> 1) * (2 * LS_REC) + 4u * h;
#pragma unroll
            for (int q = 0; q < 4; q++)
                dst[q] = h ? make_float4(blk[16 + 4 * q], blk[17 + 4 * q], blk[18 + 4 * q], blk[19 + 4 * q])
                           : make_float4(blk[4 * q], blk[4 * q + 1], blk[4 * q + 2], blk[4 * q + 3]);
        }
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

// The workgroups' shares of the first moment in a fixed order.  (Folded into ls_prep -- the last
// workgroup to finish adds them -- this cost 0.56 ms instead of saving 7 us: the release fence every
// workgroup then needs writes back its XCD's whole L2, which is full of freshly written records.)
__global__ __launch_bounds__(256) void ls_moment(const double* __restrict__ parts, uint32_t n_parts, double* __restrict__ moment) {
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

// A record in scalar registers, and its load and wait written out.  The output is early-clobber
// (it is written long after the instruction has issued, so it must not share a register with the
// address); ls_rec_wait ties the registers to the wait, so that no use can be scheduled in front
// of it.
typedef uint32_t ls_u32x16 __attribute__((ext_vector_type(16)));
struct LsRec { ls_u32x16 a, b; };        // a block of two records (see ls_prep)
__device__ __forceinline__ void ls_rec_load(LsRec& r, const float4* __restrict__ p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r.a), "=&s"(r.b) : "s"(p));
}
__device__ __forceinline__ void ls_rec_wait(LsRec& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b));
}

#ifndef LS_MIN_WAVES
#define LS_MIN_WAVES 2      // workgroups of 4 waves per CU x ... : 8 waves per SIMD fit the registers
#endif

// The waves touch blocks of two records ahead of their scalar loads (0: they do not).  With few output
// blocks per slice (one shard of eight has 6) every wave of a slice reaches a new cache line at about
// the same time and all of them wait out its way from memory; the scalar loads are only one block
// ahead.  Measured on the 1024 -> 128 cube: an eighth 0.646 -> 0.546 ms with 333 slices, 0.544 ->
// 0.526 with 424; a quarter 1.011 -> 0.984; the whole cube unchanged.
#ifndef LS_PREFETCH
#define LS_PREFETCH 1
#endif
// COUNT: the wavefronts also count the records they walk (what the arithmetic executed is priced by).
template <bool COUNT>
__global__ __launch_bounds__(LS_THREADS, LS_MIN_WAVES) void ls_pairs(
    const float4* __restrict__ rec, uint32_t n_oct_in, uint32_t slice_len, uint32_t n_slices,
    uint32_t out_size, const uint32_t* __restrict__ block_list, uint32_t oct_begin, uint32_t oct_end,
    uint32_t out_blocks, uint32_t xcd_affine, uint32_t out_begin, uint32_t out_count,
    float* __restrict__ partial, unsigned long long* __restrict__ counts) {
    const uint32_t tid = threadIdx.x;
    // Workgroups go to the 8 XCDs round-robin by their linear index and every XCD has its own
    // L2.  With xcd_affine XCD k only ever works on slices k, k + 8, ...: every input record is
    // then fetched into one L2 instead of all eight (the grid is padded to a multiple of 8 slices;
    // workgroups of the padding leave at once).
    uint32_t slice, ob;
    if (xcd_affine) {
        const uint32_t seq = blockIdx.x >> 3;
        slice = (blockIdx.x & 7u) + 8u * (seq / out_blocks);
        ob = seq % out_blocks;
        if (slice >= n_slices) return;
    } else {
        slice = blockIdx.x / out_blocks;
        ob = blockIdx.x % out_blocks;
    }
    const uint32_t lo = ob * LS_THREADS + tid;                       // local output octet
    const uint32_t oct_local = out_blocks * LS_THREADS;
    const uint32_t hw = out_size / 2;
    const uint32_t oo_raw = (block_list ? block_list[ob] * LS_THREADS : oct_begin + ob * LS_THREADS) + tid;
    const bool valid = oo_raw < oct_end;
    const uint32_t oo = valid ? oo_raw : oct_end - 1;
    uint32_t axis, oj, oi;
    octet_decode(oo, hw, ls_tile_edge(hw), axis, oj, oi);
    // Members inside the requested (face, y, x) range decide whether the wave has anything to
    // store, and nothing else: a wave that runs adds for all its octets.
    bool wanted = valid;
    if (valid && !(out_begin == 0u && out_count >= 6u * out_size * out_size)) {      // (whole cubes and shards skip this)
        wanted = false;
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) {
            Texel t = octet_member(axis, oj, oi, s, out_size, out_size);
            const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
            wanted = wanted || (o - out_begin < out_count);
        }
    }
    if (!__any(wanted)) return;          // no barrier anywhere below: a wave may leave alone

    Texel c = octet_member(axis, oj, oi, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));   // cube.cpp:97

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // FP32 sums of a chunk, per coefficient: (red, green), and blue of the even and of the odd records
    f32x2 rg[4], bb[4];
    double dacc[LS_SUMS];
#pragma unroll
    for (int q = 0; q < 4; q++) { rg[q] = f32x2{0.0f, 0.0f}; bb[q] = f32x2{0.0f, 0.0f}; }
#pragma unroll
    for (int q = 0; q < LS_SUMS; q++) dacc[q] = 0.0;

    const uint32_t in0 = slice * slice_len;                          // even: slices are whole tiles
    const uint32_t in1 = min(n_oct_in, in0 + slice_len);
    const uint32_t n_rec = in1 - in0;                                // >= 1
    const uint32_t n_blk = (n_rec + 1u) / 2u;                        // (an odd end is padded with a zero record)
    const float4* __restrict__ rp = rec + (size_t)(in0 / 2u) * (2 * LS_REC);      // wave-uniform

    const f32x2 nx = { n.x, n.x }, ny = { n.y, n.y }, nz = { n.z, n.z };
    auto eval = [&](const LsRec& r) __attribute__((always_inline)) {
        // Two records at once, one in each half of a packed instruction.
        // |x+y| + |x-y| = 2 max(|x|, |y|) and |x+y| - |x-y| = 2 sign(xy) min(|x|, |y|) give the four
        // coefficients, halved, from a+b, a-b and c with a max, a min, a max and a median:
        //   (|v0| + |v4|)/2 = max(a+b, c)      (|v0| - |v4|)/2 = min(a+b, c)
        //   (|v2| + |v1|)/2 = max(c, |a-b|)    (|v2| - |v1|)/2 = median(a-b, -c, c)
        auto fpair = [&](int at) __attribute__((always_inline)) { return f32x2{__uint_as_float(r.a[at]), __uint_as_float(r.a[at + 1])}; };
        const f32x2 gx = fpair(0), gy = fpair(2), gz = fpair(4);
        const f32x2 a = nx * gx;
        const f32x2 s1 = __builtin_elementwise_fma(ny, gy, a), s2 = __builtin_elementwise_fma(-ny, gy, a);
        const f32x2 cc = nz * gz;
        const f32x2 p = { __builtin_fmaxf(s1.x, cc.x), __builtin_fmaxf(s1.y, cc.y) };
        const f32x2 q = { __builtin_fminf(s1.x, cc.x), __builtin_fminf(s1.y, cc.y) };
        const f32x2 t = { __builtin_fmaxf(cc.x, fabsf(s2.x)), __builtin_fmaxf(cc.y, fabsf(s2.y)) };
        const f32x2 u = { __builtin_amdgcn_fmed3f(s2.x, -cc.x, cc.x), __builtin_amdgcn_fmed3f(s2.y, -cc.y, cc.y) };
        const f32x2 w[4] = { p + t, p - t, q + u, q - u };          // w^[0]/2, w^[3]/2, w^[5]/2, w^[6]/2 of both records
        // Packed multiply-adds with the record data straight from scalar registers and the record's
        // weight picked by operand selects (written out: the compiler copies a weight into a second
        // register pair for every broadcast).
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        auto spair = [&](int at) __attribute__((always_inline)) {
            return at < 16 ? u32x2{r.a[at & 15], r.a[(at + 1) & 15]} : u32x2{r.b[(at - 16) & 15], r.b[(at - 15) & 15]};
        };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(rg[k]) : "v"(w[k]), "s"(spair(8 + 6 * k)));      // first record
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(rg[k]) : "v"(w[k]), "s"(spair(10 + 6 * k)));       // second record
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bb[k]) : "v"(w[k]), "s"(spair(12 + 6 * k)));                      // blue of both
        }
    };
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dacc[3 * k] += (double)rg[k].x;
            dacc[3 * k + 1] += (double)rg[k].y;
            dacc[3 * k + 2] += (double)bb[k].x;
            dacc[3 * k + 2] += (double)bb[k].y;
            rg[k] = f32x2{0.0f, 0.0f};
            bb[k] = f32x2{0.0f, 0.0f};
        }
    };

    // Two scalar register sets (ping-pong, so nothing is copied): the next block is fetched
    // while the current one is evaluated.
    LsRec ra, rb;
    const uint32_t last = n_blk - 1u;
#if LS_PREFETCH
    // Every 64 blocks the wave touches the 64 blocks after those, a lane each, with a load nobody waits for
    // (its counter is not the scalar loads'), so that they are in the L2 when the scalar loads ask for them:
    // a load into LDS, which has no destination register that could be handed to another value while the
    // data is on its way (the words land in s_touch and are never looked at).
    __shared__ float s_touch[64];
    typedef __attribute__((address_space(1))) const void* ls_gptr;
    typedef __attribute__((address_space(3))) void* ls_lptr;
#endif
    ls_rec_load(ra, rp);
    for (uint32_t e = 0; e < n_blk; e += 2) {
#if LS_PREFETCH
        if ((e & 63u) == 0u) {
            const float4* pp = rp + (size_t)min(e + 64u + (tid & 63u), last) * (2 * LS_REC);
            __builtin_amdgcn_global_load_lds((ls_gptr)pp, (ls_lptr)s_touch, 4, 0, 0);
        }
#endif
        ls_rec_wait(ra);
        ls_rec_load(rb, rp + (size_t)min(e + 1u, last) * (2 * LS_REC));
        eval(ra);
        ls_rec_wait(rb);
        ls_rec_load(ra, rp + (size_t)min(e + 2u, last) * (2 * LS_REC));
        if (e + 1u < n_blk) eval(rb);
        if ((e & (uint32_t)(LS_CHUNK / 2 - 1)) == (uint32_t)(LS_CHUNK / 2 - 2)) flush();      // every LS_CHUNK records
    }
    ls_rec_wait(ra);       // nothing is in flight when the wave ends
#if LS_PREFETCH
    // (nothing may still be on its way into the workgroup's LDS when the wave ends)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { const float seen = s_touch[tid & 63u]; asm volatile("" : : "v"(seen)); }
#endif
    flush();

    if (valid) {
#pragma unroll
        for (int q = 0; q < LS_SUMS; q++)
            partial[((size_t)slice * LS_SUMS + q) * oct_local + lo] = (float)dacc[q];
    }
    if (COUNT && (tid & 63u) == 0u) {
        atomicAdd(&counts[3], (unsigned long long)n_rec);
        atomicAdd(&counts[4], (unsigned long long)n_rec * 4ull);      // four coefficients per record
    }
}

// The run sums from the slice sums: one thread per (four output octets, sum, run); the slices of a
// run are added in order (all eight loads are issued before the first addition: a slice missing at
// the end of the last run counts as +0).
__global__ __launch_bounds__(256) void ls_reduce(const float* __restrict__ partial, uint32_t n_slices,
                                                  uint32_t oct_local, double* __restrict__ reduced) {
    const uint32_t lo = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;      // oct_local is a multiple of 256
    const uint32_t q = blockIdx.y, g = blockIdx.z;
    if (lo >= oct_local) return;
    const uint32_t s0 = g * LS_RUN;
    float4 v[LS_RUN];
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)LS_RUN; k++)
        v[k] = s0 + k < n_slices ? *(const float4*)(partial + ((size_t)(s0 + k) * LS_SUMS + q) * oct_local + lo)
                                 : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)LS_RUN; k++) { a += (double)v[k].x; b += (double)v[k].y; c += (double)v[k].z; d += (double)v[k].w; }
    double* out = reduced + ((size_t)g * LS_SUMS + q) * oct_local + lo;
    *(double2*)out = make_double2(a, b);
    *(double2*)(out + 2) = make_double2(c, d);
}

// The runs of an output octet added up, the inverse transform, the odd half, the scale; the 8 member
// texels that lie in the requested range are stored.  One workgroup of 8 waves per 64 output octets
// (the same 64 as a wave of ls_pairs, and it leaves on the same condition): wave p adds the runs
// p, p + 8, ... in order, then the first wave adds the 8 parts in order and finishes the octets (one
// thread per octet adding all runs left this kernel with 48 workgroups for a whole 128 x 128 cube and
// 30 us).  flags[lo] is set where some texel of the octet is the difference of much larger halves;
// flags[oct_local + ob]: somewhere in its 256-octet block; flags[oct_local + out_blocks]: anywhere.
constexpr int LS_FINAL_PARTS = 8;
__global__ __launch_bounds__(64 * LS_FINAL_PARTS) void ls_final(
    const double* __restrict__ runs, uint32_t n_runs, const double* __restrict__ moment, uint32_t out_size,
    const uint32_t* __restrict__ block_list, uint32_t oct_begin, uint32_t oct_end, uint32_t out_blocks,
    uint32_t out_begin, uint32_t out_count, double scale, double limit, uint32_t* __restrict__ flags,
    unsigned long long* __restrict__ counts, float4* __restrict__ out) {
    __shared__ double s_e[LS_FINAL_PARTS - 1][LS_SUMS][64];
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t lo = blockIdx.x * 64u + lane;
    const uint32_t oct_local = out_blocks * LS_THREADS;
    const uint32_t ob = lo / LS_THREADS;
    const uint32_t oo_raw = (block_list ? block_list[ob] * LS_THREADS : oct_begin + ob * LS_THREADS) + lo % LS_THREADS;
    const bool valid = oo_raw < oct_end;
    const uint32_t oo = valid ? oo_raw : oct_end - 1;
    const uint32_t hw = out_size / 2;
    uint32_t axis, oj, oi;
    octet_decode(oo, hw, ls_tile_edge(hw), axis, oj, oi);
    {
        bool wanted = false;
        if (valid) {
#pragma unroll
            for (uint32_t s = 0; s < 8; s++) {
                Texel t = octet_member(axis, oj, oi, s, out_size, out_size);
                const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
                wanted = wanted || (o - out_begin < out_count);
            }
        }
        if (!__any(wanted)) return;          // ls_pairs stored nothing for these 64 (the same in every wave)
    }
    double e[LS_SUMS];
#pragma unroll
    for (int q = 0; q < LS_SUMS; q++) e[q] = 0.0;
    for (uint32_t b = part; b < n_runs; b += LS_FINAL_PARTS) {
#pragma unroll
        for (int q = 0; q < LS_SUMS; q++) e[q] += runs[((size_t)b * LS_SUMS + q) * oct_local + lo];
    }
    if (part > 0) {
#pragma unroll
        for (int q = 0; q < LS_SUMS; q++) s_e[part - 1][q][lane] = e[q];
    }
    __syncthreads();
    if (part > 0 || !valid) return;
    for (int p = 0; p < LS_FINAL_PARTS - 1; p++)
#pragma unroll
        for (int q = 0; q < LS_SUMS; q++) e[q] += s_e[p][q][lane];
    Texel c = octet_member(axis, oj, oi, 0, out_size, out_size);
    const F3 n = ref_normalize(ref_normal_from(c.face, c.x, c.y, out_size, out_size));
    const double nn[3] = { (double)n.x, (double)n.y, (double)n.z };
    bool bad = false;
#pragma unroll
    for (uint32_t s = 0; s < 8; s++) {
        double val[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            // (ls_pairs sums half the coefficients)
            const double even = 2.0 * (e[ch] + ls_sign(3u, s) * e[3 + ch] + ls_sign(5u, s) * e[6 + ch] + ls_sign(6u, s) * e[9 + ch]);
            const double odd = 4.0 * (ls_sign(1u, s) * nn[0] * moment[ch] + ls_sign(2u, s) * nn[1] * moment[3 + ch] +
                                      ls_sign(4u, s) * nn[2] * moment[6 + ch]);
            const double v = even + odd;               // 8 x the sum the reference forms
            bad = bad || !(fabs(even) + fabs(odd) <= limit * v);
            val[ch] = v * (scale * 0.125);
        }
        Texel t = octet_member(axis, oj, oi, s, out_size, out_size);
        const uint32_t o = (t.face * out_size + t.y) * out_size + t.x;
        if (o - out_begin < out_count) out[o] = make_float4((float)val[0], (float)val[1], (float)val[2], 0.0f);
    }
    if (bad) {
        flags[lo] = 1u;
        flags[oct_local + ob] = 1u;
        flags[oct_local + out_blocks] = 1u;
        if (counts) atomicAdd(&counts[5], 1ull);
    }
}

size_t lambertian_split_header_bytes(uint32_t n_oct_in, uint32_t out_blocks) {
    const size_t flags = ((size_t)(out_blocks * LS_THREADS + out_blocks + 1) * sizeof(uint32_t) + 255) / 256 * 256;
    const size_t parts = ((size_t)((n_oct_in + 255) / 256) * 9 * sizeof(double) + 255) / 256 * 256;
    return flags + 256 + parts;
}

size_t lambertian_split_sums_bytes(const LambertianOctPlan& pl) {
    const size_t sums = (size_t)LS_SUMS * pl.out_blocks * LS_THREADS;
    return sums * sizeof(double) * pl.n_runs + sums * sizeof(float) * pl.n_slices;
}

// The fast pass.  d_work: the header (flags | moment | moment parts), then the sums.
hipError_t lambertian_split_launch(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                   const uint32_t* d_block_list, uint32_t oct_begin, uint32_t oct_end,
                                   uint32_t out_begin, uint32_t out_count, float* d_out, void* d_rec, void* d_work,
                                   unsigned long long* d_counts, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t prep_blocks = (pl.n_oct_in + 255) / 256;
    const size_t flags_bytes = ((size_t)(pl.out_blocks * LS_THREADS + pl.out_blocks + 1) * sizeof(uint32_t) + 255) / 256 * 256;
    uint32_t* d_flags = (uint32_t*)d_work;
    double* d_moment = (double*)((char*)d_work + flags_bytes);
    double* d_parts = (double*)((char*)d_work + flags_bytes + 256);
    char* d_sums = (char*)d_work + lambertian_split_header_bytes(pl.n_oct_in, pl.out_blocks);
    const uint32_t oct_local = pl.out_blocks * LS_THREADS;
    double* d_runs = (double*)d_sums;
    float* d_slices = (float*)(d_runs + (size_t)pl.n_runs * LS_SUMS * oct_local);

    if (hipError_t e = hipMemsetAsync(d_flags, 0, flags_bytes, stream); e != hipSuccess) return e;
    ls_prep<<<prep_blocks, 256, 0, stream>>>((const float4*)d_env, in_w, pl.n_oct_in, (float4*)d_rec, d_parts);
    ls_moment<<<1, 256, 0, stream>>>(d_parts, prep_blocks, d_moment);
    const uint32_t affine = (pl.n_slices >= 8 && tuning().lamb_oct_affine != 0) ? 1u : 0u;
    const dim3 grid(pl.out_blocks * (affine ? (pl.n_slices + 7) / 8 * 8 : pl.n_slices));
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (d_counts)
        ls_pairs<true><<<grid, LS_THREADS, 0, stream>>>((const float4*)d_rec, pl.n_oct_in, pl.slice_len, pl.n_slices, out_size,
            d_block_list, oct_begin, oct_end, pl.out_blocks, affine, out_begin, out_count, d_slices, d_counts);
    else
        ls_pairs<false><<<grid, LS_THREADS, 0, stream>>>((const float4*)d_rec, pl.n_oct_in, pl.slice_len, pl.n_slices, out_size,
            d_block_list, oct_begin, oct_end, pl.out_blocks, affine, out_begin, out_count, d_slices, d_counts);
    if (ev1) (void)hipEventRecord(ev1, stream);
    ls_reduce<<<dim3((oct_local / 4 + 255) / 256, LS_SUMS, pl.n_runs), 256, 0, stream>>>(d_slices, pl.n_slices, oct_local, d_runs);
    const double scale = 4.0 * 0.31830988618379067154 / ((double)in_w * (double)in_w);      // cube.cpp:126
    ls_final<<<oct_local / 64, 64 * LS_FINAL_PARTS, 0, stream>>>(d_runs, pl.n_runs, d_moment, out_size, d_block_list, oct_begin, oct_end,
                                                pl.out_blocks, out_begin, out_count, scale,
                                                (double)tuning().lamb_split_limit, d_flags, d_counts, (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
