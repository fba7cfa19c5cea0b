// Device-side helpers shared by the IBL kernels (gfx950 only).
//
// Two kinds of arithmetic live here:
//  * "ref_*" functions reproduce the reference's float32 operation order without
//    fused multiply-add (the reference is built for baseline x86-64, which has
//    none); they are used wherever a yes/no decision or a stored direction must
//    agree bit for bit with the CPU tool (texel directions, patch cull test).
//  * everything else is free to fuse and reorder: the sums they feed are checked
//    against the reference by tolerance, not bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct F3 { float x, y, z; };

// getNormalFrom, src/utilities/cube.cpp:31-50.  s and t are evaluated in double
// and rounded once.
__device__ __forceinline__ F3 ref_normal_from(uint32_t face, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
#pragma clang fp contract(off)
    float s = (float)((((double)(float)x + 0.5) / (double)(float)w) * 2.0 - 1.0);
    float t = (float)((((double)(float)y + 0.5) / (double)(float)h) * 2.0 - 1.0);
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

// linear::dot, include/linear.hpp:100-107: left-to-right, no contraction.
__device__ __forceinline__ float ref_dot(F3 a, F3 b) {
#pragma clang fp contract(off)
    float r = a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}

// Correctly rounded float square root and quotient.  HIP's __fsqrt_rn is the native
// (1 ulp) instruction despite its name, so these go through the IEEE paths: sqrtf and
// '/' are correctly rounded when the file is built with
// -fhip-fp32-correctly-rounded-divide-sqrt (the build sets it explicitly).
__device__ __forceinline__ float ref_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ref_div(float a, float b) { return a / b; }

// linear::normalize, include/linear.hpp:93-95,114-122: (1/sqrt(v.v)) * v with a
// correctly rounded square root and division.
__device__ __forceinline__ F3 ref_normalize(F3 v) {
#pragma clang fp contract(off)
    float inv = ref_div(1.0f, ref_sqrt(ref_dot(v, v)));
    F3 r = { inv * v.x, inv * v.y, inv * v.z };
    return r;
}

__device__ __forceinline__ F3 ref_add(F3 a, F3 b) {
    F3 r = { a.x + b.x, a.y + b.y, a.z + b.z };
    return r;
}

// The cull test of cube.cpp:256-258 (patch) and the first half of :267-271
// (texel): cosHalf^2 > cutoff * |half|^2 with half = u + n.
__device__ __forceinline__ bool ref_cone_test(F3 u, F3 n, float cutoff) {
#pragma clang fp contract(off)
    F3 half = ref_add(u, n);
    float ch = ref_dot(half, n);
    float lhs = ch * ch;
    float rhs = cutoff * ref_dot(half, half);
    return lhs > rhs;
}

// The GGX denominator NoH^2*(a2-1)+1 exactly as the reference rounds it
// (cube.cpp:267,279-280 with linear.hpp:93-122): half = u + n, NoH = dot(normalize(half), n),
// every operation in float, no contraction, correctly rounded sqrt and division.
#ifndef IBL_FAST_CORE
#define IBL_FAST_CORE 1
#endif
// 1 / sqrt(x) * v with the square root and the quotient rounded as IEEE rounds them in all but a
// vanishing share of the cases, for x in [1/4, 4] (|half|^2 of two unit vectors less than 120 degrees
// apart): the hardware's 1-ulp estimates and one residual step each (Markstein), 8 instructions
// where the general correctly rounded sequences take over 20.  The lobe core repeats the
// reference's arithmetic to reproduce the drift of its rounding, which a rare last-bit difference
// of either sign does not change.
__device__ __forceinline__ F3 near_ref_normalize(F3 v) {
    float x;
    {
#pragma clang fp contract(off)
        x = ref_dot(v, v);
    }
    const float r = __builtin_amdgcn_rsqf(x);
    float y = x * r;
    y = __builtin_fmaf(__builtin_fmaf(-y, y, x), 0.5f * r, y);          // sqrt(x)
    float q = __builtin_amdgcn_rcpf(y);
    q = __builtin_fmaf(__builtin_fmaf(-y, q, 1.0f), q, q);              // 1 / sqrt(x)
    F3 o = { q * v.x, q * v.y, q * v.z };
    return o;
}

__device__ __forceinline__ float ref_ggx_denominator(F3 u, F3 n, float a2m1) {
#pragma clang fp contract(off)
    F3 half = ref_add(u, n);
    F3 hn = IBL_FAST_CORE ? near_ref_normalize(half) : ref_normalize(half);
    float noh = ref_dot(hn, n);
    float den = noh * noh;
    den = den * a2m1;
    den = den + 1.0f;
    return den;
}

// Output texel index (face, y, x order) -> unit direction, cube.cpp:97,242.
__device__ __forceinline__ F3 ref_out_normal(uint32_t o, uint32_t w, uint32_t h) {
    uint32_t f = o / (w * h);
    uint32_t rem = o - f * (w * h);
    uint32_t y = rem / w;
    uint32_t x = rem - y * w;
    return ref_normalize(ref_normal_from(f, x, y, w, h));
}

// v_fma_f32 with the clamp output modifier: max(0, min(1, a*b+c)) in one VALU
// issue (the compiler folds a median with 0 and 1 into the modifier of the instruction that
// produces its operand, and keeps the instruction's wait states after a v_sqrt / v_rcp operand,
// which it does not see through an asm statement).  Every dot product clamped this way in these kernels is between a unit
// vector and a vector of length <= 1, so the upper bound never binds and this is
// the max(0, .) of cube.cpp:112-115 for free.
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) {
    return __builtin_amdgcn_fmed3f(__builtin_fmaf(a, b, c), 0.0f, 1.0f);
}

// ---- mirror octets -------------------------------------------------------------
// The texel-centre directions of a cube map with even face size are closed under
// the 8 sign flips (+-x, +-y, +-z).  An "octet" is such a group of 8 texels; it is
// named by (axis, j, i): the pair of faces it lies on (0: +-X, 1: +-Y, 2: +-Z) and
// the row/column (j, i) < (H/2, W/2) of its member in the top-left face quadrant.
// Member sigma (bit 0/1/2 set = x/y/z component negative) of that octet is the
// texel returned here; it follows from the face table of getNormalFrom
// (cube.cpp:35-46) with s(i) < 0 for i < W/2 and s(W-1-i) = -s(i).
struct Texel { uint32_t face, x, y; };

__host__ __device__ inline Texel octet_member(uint32_t axis, uint32_t j, uint32_t i, uint32_t sigma,
                                              uint32_t w, uint32_t h) {
    const bool nx = sigma & 1u, ny = sigma & 2u, nz = sigma & 4u;
    const uint32_t xi = i, xm = w - 1 - i, yj = j, ym = h - 1 - j;
    Texel t;
    if (axis == 0) {
        t.face = nx ? 1u : 0u;
        t.y = ny ? ym : yj;
        t.x = (t.face == 0) ? (nz ? xm : xi) : (nz ? xi : xm);
    } else if (axis == 1) {
        t.face = ny ? 3u : 2u;
        t.x = nx ? xi : xm;
        t.y = (t.face == 2) ? (nz ? yj : ym) : (nz ? ym : yj);
    } else {
        t.face = nz ? 5u : 4u;
        t.y = ny ? ym : yj;
        t.x = (t.face == 4) ? (nx ? xi : xm) : (nx ? xm : xi);
    }
    return t;
}

// Octets are numbered tile-major: tiles of tw x tw canonical (top-left quadrant) positions,
// tile after tile over the three axes, so that a run of consecutive numbers is a patch of
// neighbouring directions.  hw = half the face width; tw divides hw.
__host__ __device__ inline void octet_decode(uint32_t g, uint32_t hw, uint32_t tw, uint32_t& axis, uint32_t& oj, uint32_t& oi) {
    const uint32_t t2 = tw * tw, tpr = hw / tw;
    const uint32_t tile = g / t2, within = g - tile * t2;
    const uint32_t ly = within / tw, lx = within - ly * tw;
    axis = tile / (tpr * tpr);
    const uint32_t trem = tile - axis * (tpr * tpr);
    const uint32_t ty = trem / tpr, tx = trem - ty * tpr;
    oj = ty * tw + ly;
    oi = tx * tw + lx;
}

// The number octet_decode takes apart.
__host__ __device__ inline uint32_t octet_encode(uint32_t axis, uint32_t oj, uint32_t oi, uint32_t hw, uint32_t tw) {
    const uint32_t tpr = hw / tw;
    const uint32_t tile = axis * (tpr * tpr) + (oj / tw) * tpr + oi / tw;
    return tile * (tw * tw) + (oj % tw) * tw + oi % tw;
}

// Does unit u (octets u*unit .. u*unit+unit-1 of the tile-major order) hold a texel whose
// (face, y, x) index lies in [begin, begin + count)?
__host__ __device__ inline bool octet_unit_in_range(uint32_t u, uint32_t unit, uint32_t hw, uint32_t tw, uint32_t begin,
                                                    uint32_t count) {
    const uint32_t n_oct = 3 * hw * hw, w = 2 * hw;
    for (uint32_t k = 0; k < unit; k++) {
        const uint32_t oo = u * unit + k;
        if (oo >= n_oct) break;
        uint32_t axis, oj, oi;
        octet_decode(oo, hw, tw, axis, oj, oi);
        for (uint32_t m = 0; m < 8; m++) {
            Texel t = octet_member(axis, oj, oi, m, w, w);
            if ((t.face * w + t.y) * w + t.x - begin < count) return true;     // unsigned wrap handles o < begin
        }
    }
    return false;
}

// ---- hexadecets -----------------------------------------------------------------
// The same grids are also closed under the exchange of x and y (faces +-X <-> +-Y, the Z faces
// transposed): an octet and its image under the exchange form a "hexadecet" of 16 texels (8 where the
// octet is its own image, on the diagonal of the Z faces).  Canonical hexadecets: every octet of the +-X
// faces, and the octets (j, i) with i <= j of the +-Z faces.  ggx_hex.hip works on them.
constexpr uint32_t GH_UNIT = 64;            // output hexadecets per wave: one tile
constexpr uint32_t GH_TILE = 8;             // edge of that tile in canonical positions

// q = row (row + 1) / 2 + col, col <= row
__host__ __device__ inline void tri_decode(uint32_t q, uint32_t& row, uint32_t& col) {
    uint32_t r = (uint32_t)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > q) r--;
    while ((r + 1) * (r + 2) / 2 <= q) r++;
    row = r;
    col = q - r * (r + 1) / 2;
}

// The canonical octet (axis 0: +-X faces, 2: +-Z faces; row j, column i of its member with all components
// positive) behind lane `lane` of output unit `unit`.  valid: the lane is a hexadecet of the cube (lanes
// above the diagonal of a diagonal Z tile are not); diagonal: it is its own partner.
struct HexLane { uint32_t axis, j, i; bool valid, diagonal; };
__host__ __device__ inline HexLane hex_unit_lane(uint32_t unit, uint32_t lane, uint32_t hw) {
    const uint32_t t = hw / GH_TILE;
    HexLane h;
    uint32_t tj, ti;
    if (unit < t * t) { h.axis = 0; tj = unit / t; ti = unit - tj * t; }
    else { h.axis = 2; tri_decode(unit - t * t, tj, ti); }
    h.j = tj * GH_TILE + lane / GH_TILE;
    h.i = ti * GH_TILE + lane % GH_TILE;
    h.valid = h.axis == 0 || h.i <= h.j;
    h.diagonal = h.axis == 2 && h.i == h.j;
    return h;
}
__host__ __device__ inline uint32_t hex_units_total(uint32_t out_w) {
    const uint32_t t = (out_w / 2) / GH_TILE;
    return t * t + t * (t + 1) / 2;
}
// Member m of the hexadecet of canonical octet (axis, j, i): bits 0-2 the sign flips, bit 3 the exchange.
__host__ __device__ inline Texel hex_member(uint32_t axis, uint32_t j, uint32_t i, uint32_t m, uint32_t w) {
    if (m & 8u) return octet_member(axis == 0 ? 1u : 2u, i, j, m & 7u, w, w);
    return octet_member(axis, j, i, m & 7u, w, w);
}
// Patch hexadecet ph of a patch grid with hpw patch octets per axis and row.
__host__ __device__ inline void hex_patch_decode(uint32_t ph, uint32_t hpw, uint32_t& axis, uint32_t& pj, uint32_t& pi) {
    if (ph < hpw * hpw) { axis = 0; pj = ph / hpw; pi = ph - pj * hpw; }
    else { axis = 2; tri_decode(ph - hpw * hpw, pj, pi); }
}
__host__ __device__ inline uint32_t hex_patches_total(uint32_t pw) {
    const uint32_t hpw = pw / 2;
    return hpw * hpw + hpw * (hpw + 1) / 2;
}
__host__ __device__ inline uint32_t hex_swapbits(uint32_t t) { return (t & 4u) | ((t & 1u) << 1) | ((t >> 1) & 1u); }

// ---- orbits of the cube's whole symmetry group ----------------------------------------------------
// The grids are closed under all six axis permutations as well: a texel direction (c_i, c_j, 1) reaches
// 48 texels (24 for i = j) under the sign flips and the permutations, an "orbit".  Canonical orbits: the
// octets (j, i), i <= j, of the +-Z faces, numbered q = j (j + 1) / 2 + i.  lambertian_o48.hip works on them.
struct O48Lane { uint32_t j, i; bool valid, diagonal; };
__host__ __device__ inline uint32_t o48_orbits_total(uint32_t w) { return (w / 2) * (w / 2 + 1) / 2; }
__host__ __device__ inline O48Lane o48_lane(uint32_t q, uint32_t hw) {
    O48Lane o;
    o.valid = q < hw * (hw + 1) / 2;
    tri_decode(o.valid ? q : 0u, o.j, o.i);
    o.diagonal = o.i == o.j;
    return o;
}
// The same orbits numbered tile by tile (8 x 8 canonical positions, the tiles along the lower triangle; positions
// above the diagonal of a diagonal tile are no orbits): neighbouring numbers are neighbouring texels on all six
// faces, which is what a kernel that reads the texels wants.  hw is a multiple of 8.
__host__ __device__ inline uint32_t o48_tiled_total(uint32_t w) { const uint32_t t = (w / 2) / GH_TILE; return t * (t + 1) / 2 * GH_UNIT; }
__host__ __device__ inline O48Lane o48_tiled_lane(uint32_t q, uint32_t hw) {
    const uint32_t t = hw / GH_TILE;
    O48Lane o;
    uint32_t tj, ti;
    const bool inside = q < t * (t + 1) / 2 * GH_UNIT;
    tri_decode(inside ? q / GH_UNIT : 0u, tj, ti);
    o.j = tj * GH_TILE + (q % GH_UNIT) / GH_TILE;
    o.i = ti * GH_TILE + (q % GH_UNIT) % GH_TILE;
    o.valid = inside && o.i <= o.j;
    o.diagonal = o.i == o.j;
    return o;
}
// Member m of the orbit of (j, i): bits 0-2 the sign flips, m / 8 the permutation: with n0 = (c_i, c_j, 1) the
// directions (n0x, n0y, n0z), (n0y, n0x, n0z) on the Z faces, (n0z, n0y, n0x), (n0z, n0x, n0y) on the X faces,
// (n0x, n0z, n0y), (n0y, n0z, n0x) on the Y faces (getNormalFrom's face table, cube.cpp:35-46).
__host__ __device__ inline Texel o48_member(uint32_t j, uint32_t i, uint32_t m, uint32_t w) {
    const uint32_t p = m >> 3, axis = p < 2u ? 2u : (p < 4u ? 0u : 1u);
    return (p & 1u) ? octet_member(axis, i, j, m & 7u, w, w) : octet_member(axis, j, i, m & 7u, w, w);
}

// v_add_f32 / v_sub_f32 with the clamp output modifier (result clamped to [0, 1]).
__device__ __forceinline__ float add_clamp01(float a, float b) {
    return __builtin_amdgcn_fmed3f(a + b, 0.0f, 1.0f);
}
__device__ __forceinline__ float sub_clamp01(float a, float b) {
    return __builtin_amdgcn_fmed3f(a - b, 0.0f, 1.0f);
}
__device__ __forceinline__ float mul_clamp01(float a, float b) {
    return __builtin_amdgcn_fmed3f(a * b, 0.0f, 1.0f);
}
// max(0, min(1, -a - b))
__device__ __forceinline__ float neg_sub_clamp01(float a, float b) {
    return __builtin_amdgcn_fmed3f(-a - b, 0.0f, 1.0f);
}
