// Device code shared by the mirror-symmetry GGX kernels (ggx_octets.hip, ggx_hex.hip; gfx950): the
// weight of cube.cpp:274-290 as a function of the cosine, the scalar loads of the records'
// directions, and the bf16 helpers of the matrix form.
#pragma once
#include "ibl_device.h"

namespace ibl {

constexpr int GO_THREADS = 256;         // threads per workgroup of the small kernels
#ifndef GO_WG
#define GO_WG 64                        // threads per workgroup of the pair kernels (waves are independent)
#endif
constexpr int GO_CHUNK = 256;          // most terms an FP32 chunk sum may hold
constexpr int GO_MARK_PART = 64;       // records per walk over a marked patch (the pair kernels' MARKS = 2)

typedef uint32_t go_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t go_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t go_u32x16 __attribute__((ext_vector_type(16)));

typedef float go_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 go_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int GO_MX_GROUP = 16;                       // records per matrix instruction

__device__ __forceinline__ uint32_t go_bf16_bits(float x) {          // nearest even; finite x
    const uint32_t b = __float_as_uint(x);
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

// Four records' directions (float4 each) in scalar registers (matrix form).
struct GoDir4 { go_u32x16 v; };
__device__ __forceinline__ void go_dir_load(GoDir4& r, const float4* __restrict__ p) {
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(r.v) : "s"(p));
}
__device__ __forceinline__ void go_dir_wait(GoDir4& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.v));
}
// bf16(a) in the lower half, bf16(b) in the upper half, nearest even (v_cvt_pk_bf16_f32; as a
// conversion the compiler sees, so that it keeps the wait states of the lane swap that reads it).
__device__ __forceinline__ uint32_t go_pack_bf16(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = { a, b };
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

struct GoConst {
    float cutoff, om_a2, cmin;
    float core_thr;   // below this denominator the reference's own rounding is reproduced (0: never)
};

// One record under one sign pattern: weight from the cosine, added to the sums of all 8 members.
// The pattern enters only through nt, the output direction with the pattern's signs applied
// (the cosine of output member so with input member si = so ^ tau is nt . u for every so), and
// through the bookkeeping of the caller: acc is indexed by the INPUT member, and the caller adds
// acc[si] to output member si ^ tau when it folds the chunk into the running sums.
// CORE: some lane of the wave may be inside the lobe core for this patch (see below).
template <bool COUNT, bool CORE>
__device__ __forceinline__ float go_weight(float c, float ux, float uy, float uz, float thr, float a2p, const GoConst& k,
                                           const F3& nt, uint32_t& n_acc) {
    float u = fmaf(-0.5f, c, 0.5f);
    float t = 1.0f - u;
    float sq = __builtin_amdgcn_sqrtf(t);
    float d = fmaf(u, k.om_a2, a2p);
    // At the centre of a narrow lobe d cancels to ~a2, the weight ~1/d^2 amplifies the
    // reference's float rounding of NoH^2 (about 1e-7/a2 per pair), and that rounding is
    // not zero-mean there (1/sqrt of |half|^2 just below 4 lands on rounding ties).  Inside
    // the lobe core the denominator is therefore recomputed with the reference's exact
    // operation sequence (flipping the signs of n instead of those of u flips the signs of
    // the half vector's components and changes no product, sum or rounding).  Output tiles
    // are small against the core, so the branch is nearly wave-uniform; patches too far from
    // the wave's directions to reach the core are walked without the test (CORE = false).
    if (CORE) {
        const bool core = d < k.core_thr;
        if (__any(core)) {
            F3 uu = { ux, uy, uz };
            float den = ref_ggx_denominator(uu, nt, -k.om_a2);
            d = core ? den : d;
        }
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
    return f;
}

// Four records at once (matrix form, patches away from the lobe core): the same operations in the
// same order as go_weight, the multiplies, adds and FMAs as v_pk_*_f32 on a 4-vector, which the
// compiler issues as two packed instructions per operation, the halves alternating -- so that no
// packed instruction reads the result of the one right before it (the compiler puts a wait state
// between two such, and a wait state costs an issue slot like any instruction).
// NARROW (cutoff > 1/2, i.e. the acceptance window ends above the horizon): a texel inside the
// window has a positive cosine, so the horizon clamp of c sqrt(t) has nothing to do, and
// sqrt(t) / d^2 is taken as t rsq(t d^4): one half-rate instruction per record instead of two
// (v_sqrt_f32, v_rcp_f32) and no unpacked multiply, for three more packed multiplies per pair.
// t d^4 cannot be 0 (the smallest normal float is added: d >= alpha^2 > 1e-4, so it changes no
// value above 1e-30, and at t = 0, where the window's clamp has long made the weight 0, it keeps
// rsq finite).
typedef float go_f32x4 __attribute__((ext_vector_type(4)));
template <bool NARROW>
__device__ __forceinline__ go_f32x4 go_weight_quad(go_f32x4 c, float thr, float a2p, const GoConst& k) {
    const go_f32x4 half = { 0.5f, 0.5f, 0.5f, 0.5f }, mhalf = -half;
    const go_f32x4 om = { k.om_a2, k.om_a2, k.om_a2, k.om_a2 }, a2 = { a2p, a2p, a2p, a2p };
    go_f32x4 u = __builtin_elementwise_fma(mhalf, c, half);
    go_f32x4 t = __builtin_elementwise_fma(half, c, half);      // (as 1 - u the compiler issues four unpacked subtractions)
    go_f32x4 d = __builtin_elementwise_fma(u, om, a2);
    const go_f32x4 cw = { sub_clamp01(t.x, thr), sub_clamp01(t.y, thr), sub_clamp01(t.z, thr), sub_clamp01(t.w, thr) };
    if constexpr (NARROW) {
        const float tn = 1.1754944e-38f;
        const go_f32x4 tiny = { tn, tn, tn, tn };
        const go_f32x4 d2 = d * d;
        const go_f32x4 e = __builtin_elementwise_fma(t * d2, d2, tiny);
        const go_f32x4 r = { __builtin_amdgcn_rsqf(e.x), __builtin_amdgcn_rsqf(e.y), __builtin_amdgcn_rsqf(e.z),
                             __builtin_amdgcn_rsqf(e.w) };
        return c * ((cw * cw) * (t * r));
    } else {
        const go_f32x4 sq = { __builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y), __builtin_amdgcn_sqrtf(t.z),
                              __builtin_amdgcn_sqrtf(t.w) };
        const go_f32x4 rd = { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z),
                              __builtin_amdgcn_rcpf(d.w) };
        go_f32x4 qd = cw * rd;
        const go_f32x4 m = { mul_clamp01(c.x, sq.x), mul_clamp01(c.y, sq.y), mul_clamp01(c.z, sq.z), mul_clamp01(c.w, sq.w) };
        return m * (qd * qd);
    }
}


}  // namespace ibl
