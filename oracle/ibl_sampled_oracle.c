/*
 * ibl_sampled_oracle.c -- CPU statement of the SAMPLED GGX prefilter mode.
 * TEST INFRASTRUCTURE ONLY, like ibl_oracle.c.
 *
 * This mode has NO counterpart in the reference: src/utilities/cube.cpp integrates
 * every GGX level by exhaustive quadrature over the input texels (cube.cpp:240-313).
 * The sampled mode estimates the same normalised integral,
 *       sum_l L(l) D(NoH) NoH NoL cw(l)^2  /  sum_l D(NoH) NoH NoL cw(l)^2,
 * (NoL dw: the Jacobian term of cube.cpp:274-276; D NoH: `prob`, :279-281; cw: the
 * cutoff window, :284-290) with N half vectors drawn from D(h) NoH around n = v
 * (Hammersley points; the density over l is then D/4), so that the weight of a sample
 * is NoH NoL cw^2, and reads the environment through a box-filtered
 * mip pyramid at the level whose texel matches the sample's solid angle.
 * Parity status: defined by this file; checked statistically against the quadrature
 * oracle (ibl_oracle.c), never byte-wise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float x, y, z; } s3f;

/* getNormalFrom and normalize exactly as ibl_oracle.c states them (cube.cpp:31-50,
 * linear.hpp:114-122); repeated here so that the two files stay independent. */
static s3f s_normal_from(uint32_t face, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
    float s = (float)((((double)(float)x + 0.5) / (double)(float)w) * 2.0 - 1.0);
    float t = (float)((((double)(float)y + 0.5) / (double)(float)h) * 2.0 - 1.0);
    s3f r;
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

static s3f s_face_dir(int face, float s, float t) {
    s3f r;
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

static s3f s_normalize(s3f v) {
    float d = v.x * v.x; d += v.y * v.y; d += v.z * v.z;
    float inv = 1.0f / sqrtf(d);
    s3f r = { inv * v.x, inv * v.y, inv * v.z };
    return r;
}

/* Inverse of the face table of getNormalFrom: direction -> (face, s, t). */
static void s_dir_to_face(s3f d, int* face, float* s, float* t) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) {
        float inv = 1.0f / ax;
        if (d.x > 0.0f) { *face = 0; *s = -d.z * inv; *t = -d.y * inv; }
        else            { *face = 1; *s = d.z * inv;  *t = -d.y * inv; }
    } else if (ay >= az) {
        float inv = 1.0f / ay;
        if (d.y > 0.0f) { *face = 2; *s = d.x * inv; *t = d.z * inv; }
        else            { *face = 3; *s = d.x * inv; *t = -d.z * inv; }
    } else {
        float inv = 1.0f / az;
        if (d.z > 0.0f) { *face = 4; *s = d.x * inv;  *t = -d.y * inv; }
        else            { *face = 5; *s = -d.x * inv; *t = -d.y * inv; }
    }
}

/* Number of levels of the box pyramid of a w-texel face: w, w/2, ..., down to an odd
 * size or 1. */
uint32_t orc_pyramid_levels(uint32_t w) {
    uint32_t n = 1;
    while (w > 1 && (w & 1u) == 0) { w >>= 1; n++; }
    return n;
}

/* float4 offset of pyramid level k (levels are stored back to back, level 0 first). */
uint64_t orc_pyramid_offset(uint32_t w, uint32_t level) {
    uint64_t off = 0;
    for (uint32_t k = 0; k < level; k++) off += (uint64_t)6 * (w >> k) * (w >> k);
    return off;
}

typedef struct { const float* base; uint32_t w; uint32_t levels; } pyramid_t;

/* One texel of a level; a tap outside the face is taken from the texel of the
 * neighbouring face that its direction falls on (seamless addressing). */
static void s_texel(const pyramid_t* p, uint32_t level, int face, int x, int y, float out[3]) {
    const int S = (int)(p->w >> level);
    if (x < 0 || y < 0 || x >= S || y >= S) {
        float s = ((float)x + 0.5f) / (float)S * 2.0f - 1.0f;
        float t = ((float)y + 0.5f) / (float)S * 2.0f - 1.0f;
        s3f d = s_face_dir(face, s, t);
        float s2, t2;
        s_dir_to_face(d, &face, &s2, &t2);
        x = (int)floorf((s2 + 1.0f) * 0.5f * (float)S);
        y = (int)floorf((t2 + 1.0f) * 0.5f * (float)S);
        x = x < 0 ? 0 : (x >= S ? S - 1 : x);
        y = y < 0 ? 0 : (y >= S ? S - 1 : y);
    }
    const float* q = p->base + 4 * (orc_pyramid_offset(p->w, level) + ((uint64_t)face * S + y) * S + x);
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2];
}

/* The layout the GPU library keeps the pyramid in: every face of every level with a
 * one-texel apron, apron position (x, y) (x or y equal to -1 or S) holding the texel that
 * s_texel returns for it.  Offsets in float4 texels. */
uint64_t orc_pyramid_padded_offset(uint32_t w, uint32_t level) {
    uint64_t off = 0;
    for (uint32_t k = 0; k < level; k++) off += (uint64_t)6 * ((w >> k) + 2) * ((w >> k) + 2);
    return off;
}

void orc_pyramid_pad(const float* pyramid, uint32_t w, float* out) {
    pyramid_t p = { pyramid, w, orc_pyramid_levels(w) };
    for (uint32_t level = 0; level < p.levels; level++) {
        const int S = (int)(w >> level), P = S + 2;
        float* dst = out + 4 * orc_pyramid_padded_offset(w, level);
        for (int face = 0; face < 6; face++)
            for (int y = -1; y <= S; y++)
                for (int x = -1; x <= S; x++) {
                    float c[3];
                    s_texel(&p, level, face, x, y, c);
                    float* q = dst + 4 * (((size_t)face * P + (y + 1)) * P + (x + 1));
                    q[0] = c[0]; q[1] = c[1]; q[2] = c[2];
                    /* alpha: 1 from the decoder at level 0, 0 from the box filter above it */
                    q[3] = level == 0 ? 1.0f : 0.0f;
                }
    }
}

static void s_bilinear(const pyramid_t* p, uint32_t level, s3f dir, float out[3]) {
    int face; float s, t;
    s_dir_to_face(dir, &face, &s, &t);
    const float S = (float)(p->w >> level);
    float fx = (s + 1.0f) * 0.5f * S - 0.5f, fy = (t + 1.0f) * 0.5f * S - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    float c00[3], c10[3], c01[3], c11[3];
    s_texel(p, level, face, x0, y0, c00);
    s_texel(p, level, face, x0 + 1, y0, c10);
    s_texel(p, level, face, x0, y0 + 1, c01);
    s_texel(p, level, face, x0 + 1, y0 + 1, c11);
    for (int k = 0; k < 3; k++) {
        float a = c00[k] + wx * (c10[k] - c00[k]);
        float b = c01[k] + wx * (c11[k] - c01[k]);
        out[k] = a + wy * (b - a);
    }
}

static void s_sample_env(const pyramid_t* p, s3f dir, float lod, float out[3]) {
    const float top = (float)(p->levels - 1);
    lod = lod < 0.0f ? 0.0f : (lod > top ? top : lod);
    float l0f = floorf(lod);
    uint32_t l0 = (uint32_t)l0f;
    float f = lod - l0f;
    float a[3];
    s_bilinear(p, l0, dir, a);
    if (f > 0.0f) {
        float b[3];
        s_bilinear(p, l0 + 1, dir, b);
        for (int k = 0; k < 3; k++) a[k] = a[k] + f * (b[k] - a[k]);
    }
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
}

static float s_radical_inverse(uint32_t i) {
    i = (i << 16) | (i >> 16);
    i = ((i & 0x55555555u) << 1) | ((i & 0xaaaaaaaau) >> 1);
    i = ((i & 0x33333333u) << 2) | ((i & 0xccccccccu) >> 2);
    i = ((i & 0x0f0f0f0fu) << 4) | ((i & 0xf0f0f0f0u) >> 4);
    i = ((i & 0x00ff00ffu) << 8) | ((i & 0xff00ff00u) >> 8);
    return (float)i * 2.3283064365386963e-10f;   /* 2^-32 */
}

/* One level of the sampled prefilter for output texels [out_begin, out_begin+out_count).
 * pyramid: box pyramid of the in_w-texel environment (orc_pyramid_* layout, float4
 * texels).  Level parameters as the quadrature uses them (cube.cpp:161-170). */
int orc_ggx_level_sampled(const float* pyramid, uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips,
                          uint32_t n_samples, float lod_bias, uint32_t out_begin, uint32_t out_count, float* out) {
    if (in_w == 0 || out_w == 0 || mips == 0 || n_samples == 0) return 1;
    const uint32_t n_out = out_w * out_w * 6;
    if (out_begin > n_out || out_count > n_out - out_begin) return 1;
    float r = (float)mip / (float)mips;
    float a2 = r * r * r * r;
    const float tol = 0.99f;
    float cutoff = 1 - (a2 * tol) / ((a2 - 1) * tol + 1);
    if (cutoff > 0.95f) cutoff = 0.95f;
    pyramid_t p = { pyramid, in_w, orc_pyramid_levels(in_w) };
    const float inv_n = 1.0f / (float)n_samples;
    const float texel_scale = 4.0f / ((float)in_w * (float)in_w);    /* solid angle of a texel at a face centre */
    const float pi = 3.14159265358979323846f;
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t o = out_begin; o < out_begin + out_count; o++) {
        uint32_t of = o / (out_w * out_w), oy = (o / out_w) % out_w, ox = o % out_w;
        s3f n = s_normalize(s_normal_from(of, ox, oy, out_w, out_w));
        s3f up = fabsf(n.z) < 0.999f ? (s3f){0.0f, 0.0f, 1.0f} : (s3f){1.0f, 0.0f, 0.0f};
        s3f tx = { up.y * n.z - up.z * n.y, up.z * n.x - up.x * n.z, up.x * n.y - up.y * n.x };
        tx = s_normalize(tx);
        s3f bx = { n.y * tx.z - n.z * tx.y, n.z * tx.x - n.x * tx.z, n.x * tx.y - n.y * tx.x };
        double acc[3] = {0.0, 0.0, 0.0}, wsum = 0.0;
        for (uint32_t i = 0; i < n_samples; i++) {
            float u1 = ((float)i + 0.5f) * inv_n;
            float u2 = s_radical_inverse(i);
            float cos2 = (1.0f - u2) / (1.0f + (a2 - 1.0f) * u2);
            float nol = 2.0f * cos2 - 1.0f;
            if (!(nol > 0.0f) || !(cos2 > cutoff)) continue;
            float cw = (cos2 - cutoff) / (1.0f - cutoff);
            float cos_t = sqrtf(cos2), sin_t = sqrtf(1.0f - cos2);
            float w = nol * cos_t * (cw * cw);
            float phi = 2.0f * pi * u1;
            float cx = sin_t * cosf(phi), cy = sin_t * sinf(phi);
            s3f h = { cx * tx.x + cy * bx.x + cos_t * n.x, cx * tx.y + cy * bx.y + cos_t * n.y,
                      cx * tx.z + cy * bx.z + cos_t * n.z };
            float k2 = 2.0f * cos_t;
            s3f l = { k2 * h.x - n.x, k2 * h.y - n.y, k2 * h.z - n.z };
            float d = cos2 * (a2 - 1.0f) + 1.0f;
            float pdf = a2 / (pi * d * d) * 0.25f;
            float m = fmaxf(fabsf(l.x), fmaxf(fabsf(l.y), fabsf(l.z)));
            float omega_s = inv_n / pdf;
            float omega_p = texel_scale * m * m * m;
            float lod = 0.5f * log2f(omega_s / omega_p) + lod_bias;
            float c[3];
            s_sample_env(&p, l, lod, c);
            acc[0] += (double)(w * c[0]); acc[1] += (double)(w * c[1]); acc[2] += (double)(w * c[2]);
            wsum += (double)w;
        }
        double inv = 1.0 / wsum;
        out[o * 4 + 0] = (float)(inv * acc[0]);
        out[o * 4 + 1] = (float)(inv * acc[1]);
        out[o * 4 + 2] = (float)(inv * acc[2]);
        out[o * 4 + 3] = 0.0f;
    }
    return 0;
}
