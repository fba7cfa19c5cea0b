// Parallax-corrected re-projection of a GGX mip level for a mirror-local environment (gfx950).
//
// The reference does this as a raster pass per (mip level, face) every frame
// (src/main.cpp:523-604): it draws the environment's bounding mesh, back faces only, from the
// camera reflected across the mirror plane (shaders/mirrorLocal.vert:78-87), and every fragment
// looks the pre-filtered cube up in the direction of its own position in the environment's local
// space (shaders/mirrorLocal.frag:17-19: textureLod(sourceCubemap, normalize(localFragPos),
// mipLevel)).  As a compute kernel, for a bounding mesh that is an axis-aligned box: the ray of
// every output texel leaves the reflected camera in the texel's cube direction (the pass's
// projection is the 90-degree face frustum, mirrorLocal.vert:24-76), its far intersection with
// the box is the back face the raster pass would shade, and the cube is read in that point's
// direction.  One thread per output texel; the kernel is a gather (<= 4 texels of 16 bytes in,
// 16 bytes out), bound by memory latency, not arithmetic.
//
// What is NOT the reference's: the lookup is bilinear inside the face with the edge texels
// repeated, where the hardware sampler filters across face edges, and the arithmetic of the
// hardware rasteriser and sampler is not reproduced to the bit.  Texels whose ray misses the
// box (a camera outside it) keep the pass's clear colour (0, 0, 0, 1); a hit closer than the
// pass's near plane (0.01) is clipped the same way.
#include "ibl_device.h"
#include "ibl_kernels.h"

namespace ibl {

struct MirrorArgs {
    float cam[3], bmin[3], bmax[3];
};

__global__ __launch_bounds__(256) void mirror_reproject_texels(const float4* __restrict__ src, uint32_t w, MirrorArgs a,
                                                                float4* __restrict__ out) {
#pragma clang fp contract(off)
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= 6 * w * w) return;
    const uint32_t f = o / (w * w), rem = o - f * (w * w), y = rem / w, x = rem - y * w;
    const F3 d = ref_normal_from(f, x, y, w, w);              // the face frustum's ray through the texel centre
    const float dv[3] = { d.x, d.y, d.z };
    float tnear = -3.0e38f, tfar = 3.0e38f;
    bool hit = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (dv[k] != 0.0f) {
            const float t1 = ref_div(a.bmin[k] - a.cam[k], dv[k]);
            const float t2 = ref_div(a.bmax[k] - a.cam[k], dv[k]);
            tnear = fmaxf(tnear, fminf(t1, t2));
            tfar = fminf(tfar, fmaxf(t1, t2));
        } else if (a.cam[k] < a.bmin[k] || a.cam[k] > a.bmax[k]) {
            hit = false;
        }
    }
    // back faces only: the far intersection, in front of the near plane (the ray's component
    // along the face axis is 1, so t is the view depth)
    hit = hit && tfar >= tnear && tfar > 0.01f;
    if (!hit) { out[o] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); return; }
    F3 p = { a.cam[0] + tfar * d.x, a.cam[1] + tfar * d.y, a.cam[2] + tfar * d.z };
    const F3 r = ref_normalize(p);
    // cube face selection and face coordinates (the face table of cube.cpp:35-46 inverted;
    // ties go to z, then y, as the graphics APIs define it)
    const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    uint32_t face;
    float sc, tc, ma;
    if (az >= ax && az >= ay) { face = r.z >= 0.0f ? 4u : 5u; sc = r.z >= 0.0f ? r.x : -r.x; tc = -r.y; ma = az; }
    else if (ay >= ax)        { face = r.y >= 0.0f ? 2u : 3u; sc = r.x; tc = r.y >= 0.0f ? r.z : -r.z; ma = ay; }
    else                      { face = r.x >= 0.0f ? 0u : 1u; sc = r.x >= 0.0f ? -r.z : r.z; tc = -r.y; ma = ax; }
    const float fw = (float)w;
    const float u = (ref_div(sc, ma) * 0.5f + 0.5f) * fw - 0.5f;
    const float v = (ref_div(tc, ma) * 0.5f + 0.5f) * fw - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float wu = u - fu, wv = v - fv;
    const int x0 = (int)fu, y0 = (int)fv;
    const int wi = (int)w;
    const uint32_t xa = (uint32_t)min(max(x0, 0), wi - 1), xb = (uint32_t)min(max(x0 + 1, 0), wi - 1);
    const uint32_t ya = (uint32_t)min(max(y0, 0), wi - 1), yb = (uint32_t)min(max(y0 + 1, 0), wi - 1);
    const float4 t00 = src[(face * w + ya) * w + xa], t01 = src[(face * w + ya) * w + xb];
    const float4 t10 = src[(face * w + yb) * w + xa], t11 = src[(face * w + yb) * w + xb];
    const float w00 = (1.0f - wu) * (1.0f - wv), w01 = wu * (1.0f - wv), w10 = (1.0f - wu) * wv, w11 = wu * wv;
    float4 c;
    c.x = ((w00 * t00.x + w01 * t01.x) + w10 * t10.x) + w11 * t11.x;
    c.y = ((w00 * t00.y + w01 * t01.y) + w10 * t10.y) + w11 * t11.y;
    c.z = ((w00 * t00.z + w01 * t01.z) + w10 * t10.z) + w11 * t11.z;
    c.w = ((w00 * t00.w + w01 * t01.w) + w10 * t10.w) + w11 * t11.w;
    out[o] = c;
}

hipError_t mirror_reproject_launch(const float* d_src, uint32_t w, const float cam[3], const float bmin[3],
                                   const float bmax[3], float* d_out, hipStream_t stream) {
    MirrorArgs a;
    for (int k = 0; k < 3; k++) { a.cam[k] = cam[k]; a.bmin[k] = bmin[k]; a.bmax[k] = bmax[k]; }
    const uint32_t n = 6 * w * w;
    mirror_reproject_texels<<<(n + 255) / 256, 256, 0, stream>>>((const float4*)d_src, w, a, (float4*)d_out);
    return hipGetLastError();
}

}  // namespace ibl
