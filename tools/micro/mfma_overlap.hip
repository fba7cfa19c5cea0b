// Micro-benchmark: which vector instructions issue while v_mfma_f32_32x32x16_bf16 is busy on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_overlap.hip -o tools/micro/mfma_overlap
// A loop turn is 2 independent matrix instructions (64 cycles of the matrix pipe) followed by K vector
// instructions of one kind on independent registers:
//   1 v_fma_f32   2 v_pk_fma_f32   3 v_rsq_f32   4 v_cvt_pk_bf16_f32   5 v_dot2c_f32_bf16   6 v_pk_mul_f32
//   7 v_sub_f32 (clamp)   8 v_permlane32_swap
// with 1 or 2 waves per SIMD.  Printed: cycles per loop turn and wave at the measured clock... in ns per turn.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int KIND, int K, int MFMAS>
__global__ __launch_bounds__(64) void k(float* out, const float* sc, int iters) {
    f16v c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = sc[i]; c1[i] = sc[i] + 1.0f; }
    u4 a = { 0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u }, b = { 0x3c003c00u, 0x3c013c00u, 0x3c003c00u, 0x3c003c00u };
    f2 p[16];
    float s[16];
    uint32_t w[16];
    for (int i = 0; i < 16; i++) { p[i] = f2{ sc[i] + threadIdx.x, sc[i] }; s[i] = sc[i] * 0.5f + threadIdx.x; w[i] = threadIdx.x + i; }
    const f2 m = { sc[3], sc[4] };
    for (int it = 0; it < iters; it++) {
        if (MFMAS >= 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        if (MFMAS >= 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(b), "v"(a));
#pragma unroll
        for (int i = 0; i < K; i++) {
            const int r = i & 15;
            if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[r]) : "v"(m.x), "v"(m.y));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[r]) : "v"(m), "v"(m));
            if (KIND == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(s[r]));
            if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[r]) : "v"(p[r].x), "v"(p[r].y));
            if (KIND == 5) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s[r]) : "v"(w[r]), "v"(w[(r + 1) & 15]));
            if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(m));
            if (KIND == 7) asm volatile("v_sub_f32_e64 %0, %0, %1 clamp" : "+v"(s[r]) : "v"(m.x));
            if (KIND == 8) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(w[r]), "+v"(w[(r + 8) & 15]));
        }
    }
    float r = 0.0f;
    for (int i = 0; i < 16; i++) r += c0[i] + c1[i] + p[i].x + p[i].y + s[i] + (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int K, int MFMAS>
float run(int waves_per_simd, int iters, float* out, const float* sc) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * waves_per_simd;
    k<KIND, K, MFMAS><<<blocks, 64>>>(out, sc, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND, K, MFMAS><<<blocks, 64>>>(out, sc, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / iters;        // ns per loop turn (per wave; waves of a SIMD run side by side)
}

template <int KIND>
void row(const char* name, float* out, const float* sc) {
    const int iters = 200000;
    for (int wps = 1; wps <= 2; wps++) {
        printf("%-22s %d wave(s)/SIMD | no matrix: K=8 %6.1f  K=16 %6.1f | 2 matrix + K: K=0 %6.1f  K=4 %6.1f  K=8 %6.1f  K=16 %6.1f  K=32 %6.1f ns/turn\n", name, wps,
               run<KIND, 8, 0>(wps, iters, out, sc), run<KIND, 16, 0>(wps, iters, out, sc),
               run<KIND, 0, 2>(wps, iters, out, sc), run<KIND, 4, 2>(wps, iters, out, sc), run<KIND, 8, 2>(wps, iters, out, sc),
               run<KIND, 16, 2>(wps, iters, out, sc), run<KIND, 32, 2>(wps, iters, out, sc));
        fflush(stdout);
    }
}

int main() {
    float *out, *sc;
    hipMalloc(&out, 256 * 4 * 2 * 64 * sizeof(float));
    hipMalloc(&sc, 64 * sizeof(float));
    float h[64];
    for (int i = 0; i < 64; i++) h[i] = 1.0f + i * 0.01f;
    hipMemcpy(sc, h, sizeof(h), hipMemcpyHostToDevice);
    row<1>("v_fma_f32", out, sc);
    row<2>("v_pk_fma_f32", out, sc);
    row<6>("v_pk_mul_f32", out, sc);
    row<3>("v_rsq_f32", out, sc);
    row<4>("v_cvt_pk_bf16_f32", out, sc);
    row<5>("v_dot2c_f32_bf16", out, sc);
    row<7>("v_sub_f32 clamp", out, sc);
    row<8>("v_permlane32_swap", out, sc);
    return 0;
}
