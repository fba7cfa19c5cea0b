// PNG scanline filtering of an RGBA8 image on the device: what the tool's PNG
// writer does between the RGBE encoder and the deflate stage.  The reference writes
// its strips with stb_image_write (cube.cpp:141,322), which filters every row with
// each of the five PNG filter types and keeps the one whose output has the least
// sum of absolute values read as signed bytes, the lowest type winning ties
// (ext/stb_image_write.h:1092-1174).  This kernel makes the same choice and emits
// the scanline stream a PNG's IDAT holds before compression: per row one filter-type
// byte followed by 4*w filtered bytes.
#include "ibl_kernels.h"

namespace ibl {

namespace {

constexpr int PNG_THREADS = 256;

__device__ __forceinline__ int paeth_predict(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// The byte a filter leaves for value x with neighbours a (left), b (above), c (above left).
__device__ __forceinline__ uint32_t filter_byte(int type, int x, int a, int b, int c) {
    int pred;
    switch (type) {
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: pred = paeth_predict(a, b, c); break;
        default: pred = 0; break;
    }
    return (uint32_t)(x - pred) & 0xffu;
}

__device__ __forceinline__ int signed_magnitude(uint32_t byte) {
    const int s = (int)(int8_t)byte;
    return s < 0 ? -s : s;
}

}  // namespace

// One workgroup per row.  Pixels outside the image count as zero (PNG's rule for the
// first row and the first pixel).
__global__ __launch_bounds__(PNG_THREADS) void png_filter_rows(const uint32_t* __restrict__ pixels, uint32_t w,
                                                               uint8_t* __restrict__ rows) {
    const uint32_t y = blockIdx.x;
    const uint32_t* cur = pixels + (size_t)y * w;
    const uint32_t* up = y ? cur - w : nullptr;
    __shared__ int wave_sums[PNG_THREADS / 64][5];
    __shared__ int chosen;

    int sums[5] = {0, 0, 0, 0, 0};
    for (uint32_t x = threadIdx.x; x < w; x += PNG_THREADS) {
        const uint32_t px = cur[x];
        const uint32_t pa = x ? cur[x - 1] : 0u;
        const uint32_t pb = up ? up[x] : 0u;
        const uint32_t pc = (up && x) ? up[x - 1] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = (px >> (8 * k)) & 0xff, a = (pa >> (8 * k)) & 0xff;
            const int b = (pb >> (8 * k)) & 0xff, c = (pc >> (8 * k)) & 0xff;
#pragma unroll
            for (int t = 0; t < 5; t++) sums[t] += signed_magnitude(filter_byte(t, v, a, b, c));
        }
    }
#pragma unroll
    for (int t = 0; t < 5; t++) {
        int s = sums[t];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6][t] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0, best_sum = 0x7fffffff;
        for (int t = 0; t < 5; t++) {
            int s = 0;
            for (int k = 0; k < PNG_THREADS / 64; k++) s += wave_sums[k][t];
            if (s < best_sum) { best_sum = s; best = t; }
        }
        chosen = best;
    }
    __syncthreads();
    const int type = chosen;

    uint8_t* dst = rows + (size_t)y * ((size_t)w * 4 + 1);
    if (threadIdx.x == 0) dst[0] = (uint8_t)type;
    for (uint32_t x = threadIdx.x; x < w; x += PNG_THREADS) {
        const uint32_t px = cur[x];
        const uint32_t pa = x ? cur[x - 1] : 0u;
        const uint32_t pb = up ? up[x] : 0u;
        const uint32_t pc = (up && x) ? up[x - 1] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = (px >> (8 * k)) & 0xff, a = (pa >> (8 * k)) & 0xff;
            const int b = (pb >> (8 * k)) & 0xff, c = (pc >> (8 * k)) & 0xff;
            dst[1 + 4 * (size_t)x + k] = (uint8_t)filter_byte(type, v, a, b, c);
        }
    }
}

hipError_t png_filter_rows_launch(const uint8_t* d_rgba8, uint32_t w, uint32_t h, uint8_t* d_rows, hipStream_t stream) {
    if (w && h) png_filter_rows<<<h, PNG_THREADS, 0, stream>>>((const uint32_t*)d_rgba8, w, d_rows);
    return hipGetLastError();
}

}  // namespace ibl
