// Micro-benchmark: do FP32 MFMA and FP32 VALU accumulation overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_mix.hip -o tools/micro/mfma_valu_mix
// One "record" of the Lambertian octet kernel is either 84 v_pk_fma_f32 + 11 plain VALU
// (64 output octets on the vector pipe) or 8 v_mfma_f32_32x32x2_f32 (64 output octets
// as two 32-row tiles, K = 8 sign patterns, N = 24 padded to 32).  Modes:
//   0  vector only            1  matrix only          2  both, interleaved in every wave
//   3  half the waves of a workgroup vector, the other half matrix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void valu_record(f2 (&pacc)[12], float (&t)[11], f2 a, const float (&v)[24]) {
    // 84 packed FMAs on 12 accumulator pairs (7 patterns x 12), 11 plain ops
#pragma unroll
    for (int p = 0; p < 7; p++)
#pragma unroll
        for (int i = 0; i < 12; i++) {
            f2 b = f2{v[2 * i], v[2 * i + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
    for (int i = 0; i < 11; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[i]) : "v"(a.x), "v"(v[i]));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* sc, int iters, float seed) {
    f2 pacc[12];
    float t[11], v[24];
    for (int i = 0; i < 12; i++) pacc[i] = f2{seed + i, seed - i};
    for (int i = 0; i < 11; i++) t[i] = seed * i;
    for (int i = 0; i < 24; i++) v[i] = sc[i] + threadIdx.x;
    f2 a = f2{seed * 0.5f + threadIdx.x, seed};
    f16v c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = seed + i; c1[i] = seed - i; }
    float ma = sc[threadIdx.x & 31], mb = sc[(threadIdx.x >> 1) & 31];
    const bool matrix_wave = (threadIdx.x >> 6) >= 4;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || (MODE == 3 && !matrix_wave)) {
            valu_record(pacc, t, a, v);
        } else if (MODE == 1 || (MODE == 3 && matrix_wave)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, c1, 0, 0, 0);
            }
        } else {   // MODE 2: the vector record cut into 8 parts, one MFMA before each
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, c0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 11; i++) {
                    f2 b = f2{v[2 * i], v[2 * i + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(a), "v"(b));
                }
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, c1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    f2 b = f2{v[2 * i], v[2 * i + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(a), "v"(b));
                }
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[j]) : "v"(a.x), "v"(v[j]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[j + 4]) : "v"(a.x), "v"(v[j]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[j + 7]) : "v"(a.x), "v"(v[j]));
            }
        }
    }
    float r = a.x;
    for (int i = 0; i < 12; i++) r += pacc[i].x + pacc[i].y;
    for (int i = 0; i < 11; i++) r += t[i];
    for (int i = 0; i < 16; i++) r += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
float run(int blocks, int iters, float* out, const float* sc) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 512>>>(out, sc, 10, 1.0f);
    hipEventRecord(a);
    k<MODE><<<blocks, 512>>>(out, sc, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float *out, *sc;
    hipMalloc(&out, 512 * 2048 * sizeof(float));
    hipMalloc(&sc, 64 * sizeof(float));
    std::vector<float> h(64, 0.001f);
    hipMemcpy(sc, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 4000;
    const char* names[] = {"vector only", "matrix only", "both in each wave", "half vector, half matrix waves"};
    for (int wps = 2; wps <= 4; wps += 2) {
        int blocks = 256 * wps / 2;   // 8 waves per block -> wps waves per SIMD
        float ms[4] = {run<0>(blocks, iters, out, sc), run<1>(blocks, iters, out, sc), run<2>(blocks, iters, out, sc),
                       run<3>(blocks, iters, out, sc)};
        for (int m = 0; m < 4; m++) {
            // "records" = 64-octet units of accumulation work done per SIMD
            double recs = (double)iters * wps * (m == 2 ? 2.0 : 1.0);
            double cyc = ms[m] * 1e-3 * 2.4e9;
            printf("waves/SIMD %d  %-32s %8.3f ms  -> %.0f cycles(2.4GHz) per 64-octet record per SIMD\n", wps, names[m], ms[m],
                   cyc / recs);
        }
    }
    return 0;
}
