// Micro-benchmark: issue cost of the vector instructions of the GGX matrix kernel's weight chain on
// gfx950, one instruction kind at a time, 16 independent register sets per wave, 1-4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/op_costs.hip -o tools/micro/op_costs
// Prints cycles per instruction and SIMD from event times, taking the clock as 2.4 GHz (so the
// "cycles" are nominal: the part runs at 2.1-2.35 GHz under these loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* sc, int iters, float seed) {
    f2 a[16], b[16];
    float p[16], q[16];
    unsigned u[16];
    for (int i = 0; i < 16; i++) {
        // (every vector operand depends on the lane: a wave-uniform value would live in a scalar register
        // and be copied into a vector register in front of every instruction)
        const float lane = threadIdx.x * 1e-3f;
        a[i] = f2{seed + i + lane, seed - i - lane}; b[i] = f2{0.5f + 0.001f * i + 1e-3f * lane, 0.25f + 1e-3f * lane};
        p[i] = seed + 0.01f * i + lane; q[i] = 0.999f - 1e-3f * lane; u[i] = 0x3f803f80u + i + threadIdx.x;
    }
    unsigned long long s64[16];
    for (int i = 0; i < 16; i++) {
        unsigned lo = __builtin_amdgcn_readfirstlane(__float_as_int(sc[2 * i]));
        unsigned hi = __builtin_amdgcn_readfirstlane(__float_as_int(sc[2 * i + 1]));
        s64[i] = ((unsigned long long)hi << 32) | lo;
    }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {          // v_pk_fma_f32, vector operands
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (MODE == 1) {   // v_pk_fma_f32, one scalar pair operand
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "s"(s64[i]));
            REP16(X)
#undef X
        } else if (MODE == 2) {   // v_pk_mul_f32, vector operands
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(X)
#undef X
        } else if (MODE == 3) {   // v_pk_mul_f32, scalar pair operand
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(s64[i]));
            REP16(X)
#undef X
        } else if (MODE == 4) {   // v_mul_f32 (VOP2), vector operands
#define X(i) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
            REP16(X)
#undef X
        } else if (MODE == 5) {   // v_sub_f32 with clamp (VOP3)
#define X(i) asm volatile("v_sub_f32_e64 %0, %0, %1 clamp" : "+v"(p[i]) : "v"(q[i]));
            REP16(X)
#undef X
        } else if (MODE == 6) {   // v_rsq_f32
#define X(i) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(p[i]));
            REP16(X)
#undef X
        } else if (MODE == 7) {   // v_cvt_pk_bf16_f32
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(p[i]), "v"(q[i]));
            REP16(X)
#undef X
        } else if (MODE == 8) {   // v_dot2c_f32_bf16 with a literal
#define X(i) asm volatile("v_dot2c_f32_bf16_e32 %0, 0x8000bf80, %1" : "+v"(p[i]) : "v"(u[i]));
            REP16(X)
#undef X
        } else if (MODE == 9) {   // v_permlane32_swap
#define X(i) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 8) & 15]));
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        } else if (MODE == 10) {  // v_fma_f32 (VOP3), vector operands
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "v"(q[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (MODE == 11) {  // rsq interleaved with packed multiplies 1 : 6 (the narrow weight chain's mix)
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            X(0) X(1) X(2) X(3) X(4) X(5)
            asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(p[0]));
            X(6) X(7) X(8) X(9) X(10) X(11)
            asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(p[1]));
            X(12) X(13)
#undef X
        }
    }
    float r = 0.0f;
    for (int i = 0; i < 16; i++) r += a[i].x + a[i].y + p[i] + __uint_as_float(u[i]);
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
    std::vector<float> h(64, 0.999f);
    hipMemcpy(sc, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 20000;
    const char* names[] = {"v_pk_fma_f32 vgpr", "v_pk_fma_f32 sgpr pair", "v_pk_mul_f32 vgpr", "v_pk_mul_f32 sgpr pair",
                           "v_mul_f32_e32", "v_sub_f32_e64 clamp", "v_rsq_f32", "v_cvt_pk_bf16_f32", "v_dot2c_f32_bf16 literal",
                           "v_permlane32_swap", "v_fma_f32 vgpr", "6 v_pk_mul : 1 v_rsq"};
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD
        float ms[12] = {run<0>(blocks, iters, out, sc), run<1>(blocks, iters, out, sc), run<2>(blocks, iters, out, sc),
                        run<3>(blocks, iters, out, sc), run<4>(blocks, iters, out, sc), run<5>(blocks, iters, out, sc),
                        run<6>(blocks, iters, out, sc), run<7>(blocks, iters, out, sc), run<8>(blocks, iters, out, sc),
                        run<9>(blocks, iters, out, sc), run<10>(blocks, iters, out, sc), run<11>(blocks, iters, out, sc)};
        for (int m = 0; m < 12; m++) {
            double instr_per_wave = 16.0 * iters;
            double cyc = ms[m] * 1e-3 * 2.4e9;
            printf("waves/SIMD %d  %-26s %8.3f ms  -> %.2f cycles (at 2.4 GHz) per instruction and SIMD\n", wps, names[m], ms[m],
                   cyc / (instr_per_wave * wps));
        }
    }
    return 0;
}
