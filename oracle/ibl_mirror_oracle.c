/*
 * ibl_mirror_oracle.c -- CPU statement of the mirror-local re-projection as a per-texel
 * computation.  TEST INFRASTRUCTURE ONLY (see ibl_oracle.c).
 *
 * PARITY STATUS: "parity unpinned".  The reference performs this step as a raster pass through
 * the Vulkan pipeline and the hardware cube sampler (src/main.cpp:523-604,
 * shaders/mirrorLocal.vert:78-87, shaders/mirrorLocal.frag:17-19); it holds no CPU statement,
 * fixture or output of it.  This file states the same geometry for an axis-aligned box as the
 * bounding mesh -- reflected camera, ray through the texel centre of the 90-degree face
 * frustum, far (back-face) intersection, lookup in the direction of the hit point -- with a
 * bilinear lookup inside the face; parity with the hardware sampler's filtering across face
 * edges and with its arithmetic is not claimed.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct { float x, y, z; } m3f;

/* getNormalFrom, src/utilities/cube.cpp:31-50 */
static m3f mirror_normal_from(uint32_t face, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
    float s = (float)((((double)(float)x + 0.5) / (double)(float)w) * 2.0 - 1.0);
    float t = (float)((((double)(float)y + 0.5) / (double)(float)h) * 2.0 - 1.0);
    m3f r;
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

/* shaders/mirrorLocal.vert:81-82: the camera in the environment's local space (transform is
 * the column-major 4x4 world -> local matrix), reflected across the plane z = -mirror_dist. */
void orc_mirror_reflect_camera(const float transform[16], const float camera[3], float mirror_dist, float out[3]) {
    float p[3];
    for (int r = 0; r < 3; r++) {
        float v = transform[r] * camera[0];
        v = v + transform[4 + r] * camera[1];
        v = v + transform[8 + r] * camera[2];
        v = v + transform[12 + r];
        p[r] = v;
    }
    out[0] = p[0];
    out[1] = p[1];
    out[2] = -(p[2] + mirror_dist) - mirror_dist;
}

void orc_mirror_reproject(const float* src, uint32_t w, const float cam[3], const float bmin[3], const float bmax[3],
                          float* out) {
    const uint32_t n = 6 * w * w;
#pragma omp parallel for schedule(static)
    for (uint32_t o = 0; o < n; o++) {
        const uint32_t f = o / (w * w), rem = o - f * (w * w), y = rem / w, x = rem - y * w;
        const m3f d = mirror_normal_from(f, x, y, w, w);
        const float dv[3] = { d.x, d.y, d.z };
        float tnear = -3.0e38f, tfar = 3.0e38f;
        int hit = 1;
        for (int k = 0; k < 3; k++) {
            if (dv[k] != 0.0f) {
                const float t1 = (bmin[k] - cam[k]) / dv[k];
                const float t2 = (bmax[k] - cam[k]) / dv[k];
                tnear = fmaxf(tnear, fminf(t1, t2));
                tfar = fminf(tfar, fmaxf(t1, t2));
            } else if (cam[k] < bmin[k] || cam[k] > bmax[k]) {
                hit = 0;
            }
        }
        hit = hit && tfar >= tnear && tfar > 0.01f;     /* back face in front of the near plane (mirrorLocal.vert:69-76) */
        float* c = out + (size_t)o * 4;
        if (!hit) { c[0] = c[1] = c[2] = 0.0f; c[3] = 1.0f; continue; }   /* the pass's clear colour, main.cpp:540-542 */
        m3f p = { cam[0] + tfar * d.x, cam[1] + tfar * d.y, cam[2] + tfar * d.z };
        float l2 = p.x * p.x;
        l2 = l2 + p.y * p.y;
        l2 = l2 + p.z * p.z;
        const float inv = 1.0f / sqrtf(l2);
        const m3f r = { inv * p.x, inv * p.y, inv * p.z };               /* mirrorLocal.frag:18 */
        const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
        uint32_t face;
        float sc, tc, ma;
        if (az >= ax && az >= ay) { face = r.z >= 0.0f ? 4u : 5u; sc = r.z >= 0.0f ? r.x : -r.x; tc = -r.y; ma = az; }
        else if (ay >= ax)        { face = r.y >= 0.0f ? 2u : 3u; sc = r.x; tc = r.y >= 0.0f ? r.z : -r.z; ma = ay; }
        else                      { face = r.x >= 0.0f ? 0u : 1u; sc = r.x >= 0.0f ? -r.z : r.z; tc = -r.y; ma = ax; }
        const float fw = (float)w;
        const float u = ((sc / ma) * 0.5f + 0.5f) * fw - 0.5f;
        const float v = ((tc / ma) * 0.5f + 0.5f) * fw - 0.5f;
        const float fu = floorf(u), fv = floorf(v);
        const float wu = u - fu, wv = v - fv;
        const int x0 = (int)fu, y0 = (int)fv, wi = (int)w;
        const int xa = x0 < 0 ? 0 : (x0 > wi - 1 ? wi - 1 : x0), xb = x0 + 1 < 0 ? 0 : (x0 + 1 > wi - 1 ? wi - 1 : x0 + 1);
        const int ya = y0 < 0 ? 0 : (y0 > wi - 1 ? wi - 1 : y0), yb = y0 + 1 < 0 ? 0 : (y0 + 1 > wi - 1 ? wi - 1 : y0 + 1);
        const float* t00 = src + ((size_t)(face * w + (uint32_t)ya) * w + (uint32_t)xa) * 4;
        const float* t01 = src + ((size_t)(face * w + (uint32_t)ya) * w + (uint32_t)xb) * 4;
        const float* t10 = src + ((size_t)(face * w + (uint32_t)yb) * w + (uint32_t)xa) * 4;
        const float* t11 = src + ((size_t)(face * w + (uint32_t)yb) * w + (uint32_t)xb) * 4;
        const float w00 = (1.0f - wu) * (1.0f - wv), w01 = wu * (1.0f - wv), w10 = (1.0f - wu) * wv, w11 = wu * wv;
        for (int k = 0; k < 4; k++) c[k] = ((w00 * t00[k] + w01 * t01[k]) + w10 * t10[k]) + w11 * t11[k];
    }
}
