/*
 * ibl_oracle.c -- CPU restatement of the IBL precompute path of the reference
 * cube-map tool.  TEST INFRASTRUCTURE ONLY: nothing the product ships may link,
 * load or call this file; it exists so the HIP path can be checked against an
 * independent statement of the reference's algorithm.
 *
 * PARITY STATUS
 *   Lambertian / GGX / downscale: "parity unpinned".  The reference holds no
 *   tests, fixtures or golden outputs for these functions, and its cube.cpp
 *   cannot be compiled in this image (include/util.hpp pulls in
 *   <vulkan/vulkan.h> and <GLFW/glfw3.h>, neither of which is installed).
 *   This file therefore follows the reference source op by op (every function
 *   names the lines it restates) and is pinned only by the identities the
 *   reference code documents about itself (constant environment in, the same
 *   constant out).
 *   BRDF LUT: pinned statistically against the reference's committed output
 *   include/ggx_lut.inl (the generator is seeded from std::random_device, so
 *   no run of it, the reference's own included, reproduces that file exactly).
 *   RGBE codec: restated from src/util.cpp:93-145, no fixtures exist.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -shared -fPIC.
 * All per-term arithmetic is IEEE binary32 with no fused multiply-add, and all
 * accumulators are binary64, exactly as the reference compiled for baseline
 * x86-64 behaves.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_1_PI
#define M_1_PI 0.31830988618379067154
#endif

/* ---------------------------------------------------------------- vectors */

typedef struct { float x, y, z; } v3f;

/* linear::dot, include/linear.hpp:100-107: result = 0; result += a[i]*b[i]. */
static inline float dot3(v3f a, v3f b) {
    float r = 0.0f;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}

/* linear::normalize, include/linear.hpp:93-95,114-122: (1/sqrt(dot(v,v))) * v. */
static inline v3f normalize3(v3f v) {
    float inv = 1.0f / sqrtf(dot3(v, v));
    v3f r = { inv * v.x, inv * v.y, inv * v.z };
    return r;
}

static inline v3f add3(v3f a, v3f b) {
    v3f r = { a.x + b.x, a.y + b.y, a.z + b.z };
    return r;
}

/* 2.0f * dot(v, h) * h - v as cube.cpp:372 writes it with include/linear.hpp:79-90
 * (scalar * Vec) and :51-57 (Vec - Vec); vh is dot3(v, h). */
static inline v3f reflect3(v3f v, v3f h, float vh) {
    float two_vh = 2.0f * vh;
    v3f r = { two_vh * h.x - v.x, two_vh * h.y - v.y, two_vh * h.z - v.z };
    return r;
}

/* The vector helpers above, laid open for the test that compares them bit for bit
 * with the reference's own include/linear.hpp.  a and b hold n xyz triples; out
 * receives 11 floats per pair: dot(a,b), length2(a), normalize(a), a + b,
 * 2*dot(a,b)*b - a. */
void orc_vec_ops(const float* a, const float* b, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) {
        v3f va = { a[3 * i], a[3 * i + 1], a[3 * i + 2] };
        v3f vb = { b[3 * i], b[3 * i + 1], b[3 * i + 2] };
        float* o = out + 11 * (size_t)i;
        float d = dot3(va, vb);
        v3f nn = normalize3(va), s = add3(va, vb), r = reflect3(va, vb, d);
        o[0] = d; o[1] = dot3(va, va);
        o[2] = nn.x; o[3] = nn.y; o[4] = nn.z;
        o[5] = s.x; o[6] = s.y; o[7] = s.z;
        o[8] = r.x; o[9] = r.y; o[10] = r.z;
    }
}

/* getNormalFrom, src/utilities/cube.cpp:31-50.  s and t are evaluated in
 * double (the literals 0.5, 2.0, 1.0 are double) and rounded once to float. */
static inline v3f normal_from(uint32_t face, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
    float s = (float)((((double)(float)x + 0.5) / (double)(float)w) * 2.0 - 1.0);
    float t = (float)((((double)(float)y + 0.5) / (double)(float)h) * 2.0 - 1.0);
    v3f r;
    switch (face) {
        case 0: r.x = 1.0f;  r.y = -t;    r.z = -s;    break;
        case 1: r.x = -1.0f; r.y = -t;    r.z = s;     break;
        case 2: r.x = s;     r.y = 1.0f;  r.z = t;     break;
        case 3: r.x = s;     r.y = -1.0f; r.z = -t;    break;
        case 4: r.x = s;     r.y = -t;    r.z = 1.0f;  break;
        default: r.x = -s;   r.y = -t;    r.z = -1.0f; break;
    }
    return r;
}

void orc_normal_from(uint32_t face, uint32_t x, uint32_t y, uint32_t w, uint32_t h, float out[3]) {
    v3f n = normal_from(face, x, y, w, h);
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

/* The direction table both integrals start from: getNormalFrom for every texel in
 * (face, y, x) order, normalised as at cube.cpp:97,197,242 or raw as at
 * cube.cpp:103,187.  out holds w*h*6 float4 (x, y, z, 0). */
void orc_texel_directions(uint32_t w, uint32_t h, int normalized, float* out) {
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++) {
                v3f n = normal_from(f, x, y, w, h);
                if (normalized) n = normalize3(n);
                float* o = out + 4 * (((size_t)f * h + y) * w + x);
                o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = 0.0f;
            }
}

/* ------------------------------------------------------------- Lambertian */

/* integrateLambertian's parallel loop, src/utilities/cube.cpp:95-129, for the
 * output texels [out_begin, out_begin+out_count) in (face, y, x) order.
 * `out` is the full out_size*out_size*6*4 float cube; alpha is left untouched.
 * faithful != 0 recomputes the input direction for every pair as the reference
 * does (cube.cpp:110); 0 hoists it out of the output loop -- the values are the
 * same, only the time differs. */
int orc_lambertian(const float* env, uint32_t in_w, uint32_t in_h, uint32_t out_size,
                   uint32_t out_begin, uint32_t out_count, float* out, int faithful) {
    const uint32_t n_in = in_w * in_h * 6;
    const uint32_t n_out = out_size * out_size * 6;
    if (out_begin > n_out || out_count > n_out - out_begin) return 1;
    v3f* in_normals = NULL;
    if (!faithful) {
        in_normals = (v3f*)malloc(sizeof(v3f) * (size_t)n_in);
        if (!in_normals) return 2;
        for (uint32_t f = 0; f < 6; f++)
            for (uint32_t y = 0; y < in_h; y++)
                for (uint32_t x = 0; x < in_w; x++)
                    in_normals[(f * in_h + y) * in_w + x] = normal_from(f, x, y, in_w, in_h);
    }
    const double scale = 4.0 * M_1_PI / ((double)in_w * (double)in_h);
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t o = out_begin; o < out_begin + out_count; o++) {
        uint32_t of = o / (out_size * out_size);
        uint32_t oy = (o / out_size) % out_size;
        uint32_t ox = o % out_size;
        v3f n = normalize3(normal_from(of, ox, oy, out_size, out_size));
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (uint32_t f = 0; f < 6; f++) {
            for (uint32_t y = 0; y < in_h; y++) {
                for (uint32_t x = 0; x < in_w; x++) {
                    /* cube.cpp:104 strides rows by height; faces are square. */
                    uint32_t ei = ((f * in_h + y) * in_h + x) * 4;
                    v3f w = faithful ? normal_from(f, x, y, in_w, in_h)
                                     : in_normals[(f * in_h + y) * in_w + x];
                    float jac = dot3(n, w);
                    if (jac > 0.0) {
                        float l2 = dot3(w, w);
                        jac /= (l2 * l2);
                        ax += env[ei] * jac;
                        ay += env[ei + 1] * jac;
                        az += env[ei + 2] * jac;
                    }
                }
            }
        }
        out[o * 4 + 0] = (float)(ax * scale);
        out[o * 4 + 1] = (float)(ay * scale);
        out[o * 4 + 2] = (float)(az * scale);
    }
    free(in_normals);
    return 0;
}

/* -------------------------------------------------------------------- GGX */

/* Number of GGX mip levels, src/utilities/cube.cpp:147. */
uint32_t orc_ggx_mip_levels(uint32_t top_w, uint32_t top_h, uint32_t min_size) {
    uint32_t m = top_w > top_h ? top_w : top_h;
    return (uint32_t)floor(log2((double)m) - log2((double)min_size));
}

/* Per-level constants, src/utilities/cube.cpp:161-170,175. */
void orc_ggx_params(uint32_t mip, uint32_t mips, uint32_t in_w, uint32_t in_h,
                    float* roughness, float* alpha2, float* cutoff, uint32_t* patch) {
    float r = (float)mip / (float)mips;
    float a2 = r * r * r * r;
    const float tol = 0.99f;
    float c = 1 - (a2 * tol) / ((a2 - 1) * tol + 1);
    if (c > 0.95f) c = 0.95f;
    uint32_t m = in_w < in_h ? in_w : in_h;
    uint32_t ps = m / 16 < 16u ? m / 16 : 16u;
    *roughness = r; *alpha2 = a2; *cutoff = c; *patch = ps;
}

/* One iteration of integrateGGX's level loop, src/utilities/cube.cpp:161-309,
 * for the output texels [out_begin, out_begin+out_count).  counts (optional)
 * receives {patch tests, in-patch texel tests, accepted texels}. */
int orc_ggx_level(const float* env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                  uint32_t mip, uint32_t mips, uint32_t out_begin, uint32_t out_count,
                  float* out, uint64_t* counts) {
    float roughness, a2, cutoff; uint32_t ps;
    if (in_w < 16 || in_h < 16 || mips == 0) return 1;   /* patch size 0 divides by zero in the reference */
    orc_ggx_params(mip, mips, in_w, in_h, &roughness, &a2, &cutoff, &ps);
    const uint32_t n_in = in_w * in_h * 6;
    const uint32_t n_out = out_w * out_h * 6;
    if (out_begin > n_out || out_count > n_out - out_begin) return 1;
    const uint32_t pw = in_w / ps, ph = in_h / ps;
    v3f* env_n = (v3f*)malloc(sizeof(v3f) * (size_t)n_in);
    v3f* unit_n = (v3f*)malloc(sizeof(v3f) * (size_t)n_in);
    v3f* patch_n = (v3f*)malloc(sizeof(v3f) * (size_t)(pw * ph * 6));
    if (!env_n || !unit_n || !patch_n) { free(env_n); free(unit_n); free(patch_n); return 2; }
    /* cube.cpp:183-201 */
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < in_h; y++)
            for (uint32_t x = 0; x < in_w; x++) {
                uint32_t i = (f * in_h + y) * in_w + x;
                env_n[i] = normal_from(f, x, y, in_w, in_h);
                unit_n[i] = normalize3(env_n[i]);
            }
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < ph; y++)
            for (uint32_t x = 0; x < pw; x++)
                patch_n[(f * ph + y) * pw + x] = normalize3(normal_from(f, x, y, pw, ph));

    uint64_t c_patch = 0, c_tested = 0, c_accept = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : c_patch, c_tested, c_accept)
    for (uint32_t o = out_begin; o < out_begin + out_count; o++) {
        uint32_t of = o / (out_w * out_h);
        uint32_t oy = (o / out_w) % out_h;
        uint32_t ox = o % out_w;
        v3f n = normalize3(normal_from(of, ox, oy, out_w, out_h));
        double sx = 0.0, sy = 0.0, sz = 0.0, total = 0.0;
        for (uint32_t f = 0; f < 6; f++) {
            for (uint32_t py = 0; py < ph; py++) {
                for (uint32_t px = 0; px < pw; px++) {
                    uint32_t pi = (f * ph + py) * pw + px;
                    v3f half = add3(patch_n[pi], n);
                    float cos_half = dot3(half, n);
                    c_patch++;
                    if (!(cos_half * cos_half > cutoff * dot3(half, half))) continue;
                    for (uint32_t sy_ = 0; sy_ < ps; sy_++) {
                        for (uint32_t sx_ = 0; sx_ < ps; sx_++) {
                            uint32_t tx = px * ps + sx_;
                            uint32_t ty = py * ps + sy_;
                            /* cube.cpp:264: the row stride is the height */
                            uint32_t ei = ((f * in_h + ty) * in_h + tx) * 4;
                            v3f u = unit_n[ei / 4];
                            half = add3(u, n);
                            cos_half = dot3(half, n);
                            float cos2_half = cos_half * cos_half;
                            float cos_refl = dot3(u, n);
                            c_tested++;
                            if (cos2_half > cutoff * dot3(half, half) && cos_refl > 0.0) {
                                c_accept++;
                                v3f w = env_n[ei / 4];
                                float jac = dot3(n, w);
                                float l2 = dot3(w, w);
                                jac /= (l2 * l2);
                                float noh = dot3(normalize3(half), n);
                                float denom = noh * noh * (a2 - 1) + 1;
                                float prob = (float)((double)(a2 * noh) / (M_PI * (double)denom * (double)denom));
                                jac *= prob;
                                float cw = (float)((double)(cos2_half / dot3(half, half) - cutoff) / (1.0 - (double)cutoff));
                                jac *= cw * cw;
                                total += jac;
                                sx += env[ei] * jac;
                                sy += env[ei + 1] * jac;
                                sz += env[ei + 2] * jac;
                            }
                        }
                    }
                }
            }
        }
        /* cube.cpp:305 with linear.hpp:93-95: multiply by the reciprocal */
        double inv = 1.0 / total;
        out[o * 4 + 0] = (float)(inv * sx);
        out[o * 4 + 1] = (float)(inv * sy);
        out[o * 4 + 2] = (float)(inv * sz);
    }
    if (counts) { counts[0] = c_patch; counts[1] = c_tested; counts[2] = c_accept; }
    free(env_n); free(unit_n); free(patch_n);
    return 0;
}

/* --------------------------------------------------------------- BRDF LUT */

/* std::mt19937 as the reference instantiates it (cube.cpp:343). */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* The n-th (1-based) output of std::mt19937(seed); [rand.predef] fixes the
 * 10000th output of the default seed 5489 at 4123659995. */
uint32_t orc_mt19937_nth(uint32_t seed, uint32_t n) {
    mt19937_t g;
    mt_seed(&g, seed);
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; i++) v = mt_next(&g);
    return v;
}

/* libstdc++ uniform_real_distribution<double>(0, b) over mt19937: two 32-bit
 * draws, low word first, (lo + hi*2^32) / 2^64, then scaled (cube.cpp:344). */
static double mt_uniform(mt19937_t* g, double b) {
    double sum = (double)mt_next(g);
    sum += (double)mt_next(g) * 4294967296.0;
    double c = sum / 18446744073709551616.0;
    if (c >= 1.0) c = nextafter(1.0, 0.0);
    return b * c + 0.0;
}

/* The counter-based jitter shared with the HIP kernel: splitmix64 finaliser of
 * (seed, cell, sample); the two 32-bit halves give the two offsets in [0,1). */
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

void orc_lut_jitter(uint64_t seed, uint32_t cell, uint32_t sample, double inv_s, double* j1, double* j2) {
    uint64_t h = mix64(seed + 0x9e3779b97f4a7c15ull * ((((uint64_t)cell) << 32 | (uint64_t)sample) + 1ull));
    *j1 = (double)(uint32_t)(h & 0xffffffffu) * (1.0 / 4294967296.0) * inv_s;
    *j2 = (double)(uint32_t)(h >> 32) * (1.0 / 4294967296.0) * inv_s;
}

/* One stratified sample of generateGGXLUT, src/utilities/cube.cpp:366-380.
 * (ux, uy) is the jittered point already rounded to float (cube.cpp:366). */
static inline void lut_sample(float ux, float uy, float alpha, float a2, v3f v, double* slope, double* bias) {
    float phi = (float)(2 * M_PI * (double)ux);
    float cos_t = (float)sqrt((1.0 - (double)uy) / (1.0 + ((double)a2 - 1.0) * (double)uy));
    float sin_t = (float)sqrt(1.0 - (double)(cos_t * cos_t));
    v3f h = { (float)((double)sin_t * cos((double)phi)), (float)((double)sin_t * sin((double)phi)), cos_t };
    float vh = dot3(v, h);
    v3f l = reflect3(v, h, vh);
    if (l.z > 0.0) {
        float k = alpha / 2;
        float g1l = l.z / (l.z * (1 - k) + k);
        float g1v = v.z / (v.z * (1 - k) + k);
        float g = g1l * g1v;
        float fres = powf(1 - vh, 5);
        float norm = vh / (h.z * v.z);
        *slope += g * (1 - fres) * norm;
        *bias += g * fres * norm;
    }
}

/* generateGGXLUT, src/utilities/cube.cpp:346-385, minus the printing.
 * mode 0: std::mt19937(seed) consumed in the reference's order from cell 0
 *         (single thread; cell_begin must be 0) -- "the reference with a fixed
 *         seed";
 * mode 1: counter-based jitter, identical to the HIP kernel's;
 * mode 2: stratum midpoints (no randomness), the converged-integral mode.
 * out holds n_rough*n_cos*2 doubles, row = roughness, col = cos(theta_v). */
int orc_brdf_lut(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int mode,
                 uint32_t cell_begin, uint32_t cell_count, double* out) {
    const uint32_t n_cells = n_rough * n_cos;
    if (cell_begin > n_cells || cell_count > n_cells - cell_begin || samples_sqrt == 0) return 1;
    if (mode == 0 && cell_begin != 0) return 1;
    const double inv_s = 1.0 / (double)samples_sqrt;
    mt19937_t gen;
    if (mode == 0) mt_seed(&gen, (uint32_t)seed);
#pragma omp parallel for schedule(dynamic, 4) if (mode != 0)
    for (uint32_t cell = cell_begin; cell < cell_begin + cell_count; cell++) {
        uint32_t ri = cell / n_cos, ci = cell % n_cos;
        float roughness = (float)(((double)(float)ri + 0.5) / (double)n_rough);
        float alpha = roughness * roughness;
        float a2 = alpha * alpha;
        float cos_v = (float)(((double)(float)ci + 0.5) / (double)n_cos);
        v3f v = { (float)sqrt(1.0 - (double)(cos_v * cos_v)), 0.0f, cos_v };
        double slope = 0.0, bias = 0.0;
        for (uint32_t sx = 0; sx < samples_sqrt; sx++) {
            for (uint32_t sy = 0; sy < samples_sqrt; sy++) {
                double j1, j2;
                if (mode == 0) { j1 = mt_uniform(&gen, inv_s); j2 = mt_uniform(&gen, inv_s); }
                else if (mode == 1) orc_lut_jitter(seed, cell, sx * samples_sqrt + sy, inv_s, &j1, &j2);
                else { j1 = 0.5 * inv_s; j2 = 0.5 * inv_s; }
                float ux = (float)((double)((float)sx / (float)samples_sqrt) + j1);
                float uy = (float)((double)((float)sy / (float)samples_sqrt) + j2);
                lut_sample(ux, uy, alpha, a2, v, &slope, &bias);
            }
        }
        slope /= (double)(samples_sqrt * samples_sqrt);
        bias /= (double)(samples_sqrt * samples_sqrt);
        out[cell * 2 + 0] = slope;
        out[cell * 2 + 1] = bias;
    }
    return 0;
}

/* ------------------------------------------------------------- RGBE codec */

/* convertRGBEtoRGB, src/util.cpp:93-111. */
void orc_rgbe_decode(const uint8_t* src, float* dst, uint32_t n) {
    for (uint32_t p = 0; p < n; p++) {
        const uint8_t* s = src + (size_t)p * 4;
        float* d = dst + (size_t)p * 4;
        d[3] = 1.0f;
        if (s[0] == 0 && s[1] == 0 && s[2] == 0 && s[3] == 0) {
            d[0] = 0.0f; d[1] = 0.0f; d[2] = 0.0f;
        } else {
            int e = (int)s[3] - 128;
            d[0] = ldexpf((s[0] + 0.5f) / 256.0f, e);
            d[1] = ldexpf((s[1] + 0.5f) / 256.0f, e);
            d[2] = ldexpf((s[2] + 0.5f) / 256.0f, e);
        }
    }
}

static inline uint8_t clamp_byte(int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* convertRGBtoRGBE, src/util.cpp:113-145. */
void orc_rgbe_encode(const float* src, uint8_t* dst, uint32_t n) {
    for (uint32_t p = 0; p < n; p++) {
        const float* s = src + (size_t)p * 4;
        uint8_t* d = dst + (size_t)p * 4;
        float m = (s[0] < s[1]) ? s[1] : s[0];   /* std::max(a,b) = (a<b)?b:a */
        m = (m < s[2]) ? s[2] : m;
        if (m <= 1e-32f) { d[0] = d[1] = d[2] = d[3] = 0; continue; }
        int e;
        float frac_mult = 256.0f * frexpf(m, &e) / m;
        if (e > 127) { d[0] = d[1] = d[2] = d[3] = 255; continue; }
        d[0] = clamp_byte((int32_t)(s[0] * frac_mult));
        d[1] = clamp_byte((int32_t)(s[1] * frac_mult));
        d[2] = clamp_byte((int32_t)(s[2] * frac_mult));
        d[3] = (uint8_t)(e + 128);
    }
}

/* -------------------------------------------------------------- downscale */

/* The box filter of main, src/utilities/cube.cpp:482-501: rows run over the
 * whole 6*h strip, sums are float in (avgY, avgX) order, then one multiply by
 * 1/D^2 (Vec / scalar is reciprocal-multiply, linear.hpp:93-95).  Alpha of the
 * output is 0 (value-initialised vector, never written). */
void orc_downscale(const float* env, uint32_t w, uint32_t h, uint32_t d, float* out) {
    const uint32_t ow = w / d, oh = h / d;
    const float inv = 1.0f / (float)(d * d);
    for (uint32_t y = 0; y < oh * 6; y++) {
        for (uint32_t x = 0; x < ow; x++) {
            float ax = 0.0f, ay = 0.0f, az = 0.0f;
            for (uint32_t ay_ = 0; ay_ < d; ay_++) {
                for (uint32_t ax_ = 0; ax_ < d; ax_++) {
                    size_t fi = ((size_t)(y * d + ay_) * w + (x * d + ax_)) * 4;
                    ax += env[fi]; ay += env[fi + 1]; az += env[fi + 2];
                }
            }
            size_t oi = ((size_t)y * ow + x) * 4;
            out[oi] = inv * ax; out[oi + 1] = inv * ay; out[oi + 2] = inv * az; out[oi + 3] = 0.0f;
        }
    }
}

/* ------------------------------------------------------- PNG scanline filter */

static inline int paeth_predict(int a, int b, int c) {
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

static uint8_t png_filter_byte(int type, const uint8_t* cur, const uint8_t* up, size_t i) {
    int a = i >= 4 ? cur[i - 4] : 0, b = up ? up[i] : 0, c = (up && i >= 4) ? up[i - 4] : 0, pred = 0;
    if (type == 1) pred = a;
    else if (type == 2) pred = b;
    else if (type == 3) pred = (a + b) >> 1;
    else if (type == 4) pred = paeth_predict(a, b, c);
    return (uint8_t)(cur[i] - pred);
}

/* The filter stage of the PNG writer the tool saves its strips with (cube.cpp:141,322:
 * stb_image_write with its default settings, ext/stb_image_write.h:1092-1174): per row,
 * all five PNG filters are tried on the RGBA8 bytes (neighbours outside the image are 0,
 * which is what that file's first-row table amounts to), the one with the least sum of
 * |signed byte| is kept, and the lowest type wins ties.  rows receives (4w+1)*h bytes:
 * filter type, then the filtered row. */
void orc_png_filter_rows(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* rows) {
    const size_t stride = (size_t)w * 4;
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* cur = rgba + stride * y;
        const uint8_t* up = y ? cur - stride : NULL;
        int best = 0;
        long best_sum = -1;
        for (int t = 0; t < 5; t++) {
            long sum = 0;
            for (size_t i = 0; i < stride; i++) sum += abs((int)(int8_t)png_filter_byte(t, cur, up, i));
            if (best_sum < 0 || sum < best_sum) { best_sum = sum; best = t; }
        }
        uint8_t* dst = rows + (stride + 1) * y;
        dst[0] = (uint8_t)best;
        for (size_t i = 0; i < stride; i++) dst[1 + i] = png_filter_byte(best, cur, up, i);
    }
}
