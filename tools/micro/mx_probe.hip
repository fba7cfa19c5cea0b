// Probe: operand order of v_cvt_pk_bf16_f32, lane movement of v_permlane32_swap, and the
// operand / result layouts of v_mfma_f32_32x32x16_bf16 (gfx950).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mx_probe.hip -o tools/micro/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void probe(uint32_t* out, float* d) {
    const uint32_t lane = threadIdx.x;
    uint32_t p;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(1.0f), "v"(2.0f));
    out[lane] = p;
    auto s = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    out[64 + lane] = s[0];
    out[128 + lane] = s[1];
    // A[row][k] = 1 for (row 5, k 11) only; B[k][col] = (k + 1) * 0.5 for col < 8, element order as the guide says
    u4 a = {0, 0, 0, 0}, b;
    const uint32_t r = lane & 31, h = lane >> 5;
    if (r == 5 && h == 1) a[1] = 0x3f80u << 16;            // element j = 3: upper half of dword 1
    uint32_t e[8];
    for (int j = 0; j < 8; j++) {
        float v = r < 8 ? (8 * h + j + 1) * 0.5f : 0.0f;
        e[j] = __float_as_uint(v) >> 16;
    }
    for (int q = 0; q < 4; q++) b[q] = e[2 * q] | (e[2 * q + 1] << 16);
    f16v c;
    for (int i = 0; i < 16; i++) c[i] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    for (int i = 0; i < 16; i++) d[lane * 16 + i] = c[i];
}

int main() {
    uint32_t* out; float* d;
    hipMalloc(&out, 192 * 4); hipMalloc(&d, 64 * 16 * 4);
    probe<<<1, 64>>>(out, d);
    uint32_t h[192]; float hd[1024];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    printf("cvt_pk(1.0, 2.0) = 0x%08x  (0x40003f80: first operand in the lower half)\n", h[0]);
    printf("swap(lane, 100 + lane): first result lanes 0,31,32,63 = %u %u %u %u ; second = %u %u %u %u\n",
           h[64], h[64 + 31], h[64 + 32], h[64 + 63], h[128], h[128 + 31], h[128 + 32], h[128 + 63]);
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 16; i++)
            if (hd[l * 16 + i] != 0.0f) printf("D lane %d reg %d = %g (expect row 5 -> lanes 0-7 (cols), reg 5 %% 4 + 4 * (5 / 8).. value 6)\n", l, i, hd[l * 16 + i]);
    return 0;
}
