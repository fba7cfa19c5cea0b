// Micro-benchmark: what does one (record, sign pattern) of the octet kernels cost per SIMD if the
// accumulation runs on the matrix pipe as exact 3 x bf16 splits while the vector pipe evaluates and
// splits the weights?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3_mix.hip -o tools/micro/bf16x3_mix
// Per group of 16 (record, pattern) pairs and 64 output octets: VALU_PER weight-side vector
// instructions per pair, 3 LDS reads of 16 bytes per lane (the three bf16 pieces of the texel tile),
// TERMS x 2 v_mfma_f32_32x32x16_bf16 (TERMS = 9: every partial product; 6: those above 2^-24).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int TERMS, int VALU_PER, bool MATRIX>
__global__ __launch_bounds__(256) void k(float* out, const float* sc, int iters, float seed) {
    __shared__ u4 tile[3][64 * 4];
    for (int i = threadIdx.x; i < 3 * 64 * 4; i += blockDim.x) ((u4*)tile)[i] = u4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f16v c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = seed + i; c1[i] = seed - i; }
    float w[16];
    for (int i = 0; i < 16; i++) w[i] = sc[i] + threadIdx.x * 1e-3f;
    const float a = sc[threadIdx.x & 31], b = sc[(threadIdx.x >> 1) & 31];
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        // weight side: VALU_PER plain vector instructions per pair
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int j = 0; j < VALU_PER; j++) {
                if (j % 3 == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
                else if (j % 3 == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(w[i]));
                else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            }
        if (MATRIX) {
            u4 p[3];
#pragma unroll
            for (int q = 0; q < 3; q++) p[q] = tile[q][((lane + it) & 63) * 4 + (lane >> 5)];
            u4 wa[3];
#pragma unroll
            for (int q = 0; q < 3; q++)
                wa[q] = u4{__float_as_uint(w[4 * q]), __float_as_uint(w[4 * q + 1]), __float_as_uint(w[4 * q + 2]), __float_as_uint(w[4 * q + 3])};
#pragma unroll
            for (int t = 0; t < TERMS; t++) {
                const int qa = t % 3, qb = t / 3;
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, wa[qa]), __builtin_bit_cast(bf8, p[qb]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, wa[qb]), __builtin_bit_cast(bf8, p[qa]), c1, 0, 0, 0);
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += c0[i] + c1[i] + w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int TERMS, int VALU_PER, bool MATRIX>
void run(int wps, float* out, const float* sc) {
    const int iters = 2000, blocks = 256 * wps;          // 4 waves per block, one per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<TERMS, VALU_PER, MATRIX><<<blocks, 256>>>(out, sc, 10, 1.0f);
    hipEventRecord(a);
    k<TERMS, VALU_PER, MATRIX><<<blocks, 256>>>(out, sc, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double pairs = (double)iters * 16 * wps;        // per SIMD
    printf("waves/SIMD %d  terms %d  vector instr/pair %2d  matrix %d : %8.3f ms  %.1f ns = %.0f cycles(2.4GHz) per (record, pattern) per SIMD\n",
           wps, TERMS, VALU_PER, (int)MATRIX, ms, ms * 1e6 / pairs, ms * 1e-3 * 2.4e9 / pairs);
}

int main() {
    float *out, *sc;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipMalloc(&sc, 64 * sizeof(float));
    std::vector<float> h(64, 0.001f);
    hipMemcpy(sc, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    for (int wps = 2; wps <= 4; wps += 2) {
        run<9, 0, true>(wps, out, sc);
        run<6, 0, true>(wps, out, sc);
        run<9, 8, false>(wps, out, sc);
        run<9, 8, true>(wps, out, sc);
        run<6, 8, true>(wps, out, sc);
        run<9, 12, true>(wps, out, sc);
        run<9, 20, false>(wps, out, sc);
        run<9, 20, true>(wps, out, sc);
        run<6, 20, true>(wps, out, sc);
    }
    return 0;
}
