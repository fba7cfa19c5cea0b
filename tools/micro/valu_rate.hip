// Micro-benchmark: issue rate of FP32 FMA forms on gfx950 at 1-4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
// Each wave runs ITER iterations of 32 independent accumulator updates (the shape of
// the GGX/Lambertian accumulation), in one of four forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* sc, int iters, float seed) {
    float acc[32];
    f2 pacc[16];
    for (int i = 0; i < 32; i++) acc[i] = seed + i;
    for (int i = 0; i < 16; i++) pacc[i] = f2{seed + i, seed - i};
    float f = seed * 0.5f + threadIdx.x;
    float s[32];
    for (int i = 0; i < 32; i++) s[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc[i])));   // scalar registers
    float v[32];
    for (int i = 0; i < 32; i++) v[i] = sc[i] + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(f), "v"(v[i]));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(f), "s"(s[i]));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                f2 b = f2{v[2 * i], v[2 * i + 1]};
                f2 a = f2{f, f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(a), "v"(b));
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                f2 a = f2{f, f};
                unsigned lo = __builtin_amdgcn_readfirstlane(__float_as_int(s[2 * i]));
                unsigned hi = __builtin_amdgcn_readfirstlane(__float_as_int(s[2 * i + 1]));
                unsigned long long sb = ((unsigned long long)hi << 32) | lo;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(a), "s"(sb));
            }
        } else if (MODE == 5) {   // VOP2 encoding, vgpr operands
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(f), "v"(v[i]));
        } else if (MODE == 6) {   // VOP2 encoding, sgpr src0
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(s[i]), "v"(f));
        } else if (MODE == 4) {   // 16 plain + transcendental pair per 16, like the weight chain
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(f), "v"(v[i]));
            asm volatile("v_sqrt_f32 %0, %0\n v_rcp_f32 %0, %0" : "+v"(f));
        }
    }
    float r = f;
    for (int i = 0; i < 32; i++) r += acc[i];
    for (int i = 0; i < 16; i++) r += pacc[i].x + pacc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
float run(int blocks, int iters, float* out, const float* sc) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, sc, 10, 1.0f);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, sc, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float *out, *sc;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipMalloc(&sc, 64 * sizeof(float));
    std::vector<float> h(64, 0.001f);
    hipMemcpy(sc, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 20000;
    const char* names[] = {"v_fma vgpr", "v_fma sgpr", "v_pk_fma vgpr", "v_pk_fma sgpr", "v_fma + sqrt,rcp", "v_fmac_e32 vgpr", "v_fmac_e32 sgpr"};
    for (int wps = 1; wps <= 4; wps++) {
        int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD if one block-set per CU
        float ms[7] = {run<0>(blocks, iters, out, sc), run<1>(blocks, iters, out, sc), run<2>(blocks, iters, out, sc),
                       run<3>(blocks, iters, out, sc), run<4>(blocks, iters, out, sc), run<5>(blocks, iters, out, sc),
                       run<6>(blocks, iters, out, sc)};
        for (int m = 0; m < 7; m++) {
            double fma_per_wave = 32.0 * iters;                       // 64-lane FMAs (pk counts 2)
            double waves_per_simd = wps;
            double cyc = ms[m] * 1e-3 * 2.4e9;                        // at nominal clock
            printf("waves/SIMD %d  %-18s %8.3f ms  -> %.2f cycles(2.4GHz) per 64-lane FMA per SIMD, %.1f TFLOP/s\n", wps,
                   names[m], ms[m], cyc / (fma_per_wave * waves_per_simd),
                   2.0 * 64 * fma_per_wave * blocks * 4 / (ms[m] * 1e-3) / 1e12);
        }
    }
    return 0;
}
