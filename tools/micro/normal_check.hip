// Writes ref_normalize(ref_normal_from(face, x, y, w, w)) for every texel of a w-cube
// to a raw float file, for a bit-for-bit comparison with the CPU oracle.
//   hipcc --offload-arch=gfx950 -O3 -I../../vulkan-renderer_amd/csrc normal_check.hip -o normal_check
#include "ibl_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void normals(uint32_t w, float* out) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= 6 * w * w) return;
    F3 n = ref_out_normal(o, w, w);
    out[3 * o] = n.x; out[3 * o + 1] = n.y; out[3 * o + 2] = n.z;
}

int main(int argc, char** argv) {
    uint32_t w = argc > 1 ? (uint32_t)atoi(argv[1]) : 256;
    const char* path = argc > 2 ? argv[2] : "normals.bin";
    size_t n = (size_t)6 * w * w * 3;
    float* d;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) return 1;
    normals<<<(6 * w * w + 255) / 256, 256>>>(w, d);
    std::vector<float> h(n);
    if (hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    FILE* f = fopen(path, "wb");
    if (!f) return 3;
    fwrite(h.data(), sizeof(float), n, f);
    fclose(f);
    hipFree(d);
    return 0;
}
