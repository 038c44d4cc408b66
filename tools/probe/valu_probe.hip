// Micro-probe: issue cost of VALU instruction types on one SIMD, alone and beside a partner wave streaming MFMAs.
// 512 threads = 8 waves = 2 per SIMD; waves 0-3 run the VALU stream, waves 4-7 the MFMA stream (or idle).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP8(x) x x x x x x x x
template <int KIND, int MFMA>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, float* sink, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 1e-3f + i); fb[i] = (_Float16)(i * 0.5f); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) { REP8(asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (KIND == 1) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (KIND == 2) { REP8(asm volatile("v_cvt_pk_f16_f32 %8, %0, %1\n v_cvt_pk_f16_f32 %9, %2, %3\n v_cvt_pk_f16_f32 %10, %4, %5\n v_cvt_pk_f16_f32 %11, %6, %7\n v_cvt_pk_f16_f32 %8, %1, %0\n v_cvt_pk_f16_f32 %9, %3, %2\n v_cvt_pk_f16_f32 %10, %5, %4\n v_cvt_pk_f16_f32 %11, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
            if (KIND == 3) { REP8(asm volatile("v_fma_mixlo_f16 %8, -%8, 1.0, %0 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %9, -%9, 1.0, %1 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %10, -%10, 1.0, %2 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %11, -%11, 1.0, %3 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %8, -%8, 1.0, %4 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %9, -%9, 1.0, %5 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %10, -%10, 1.0, %6 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %11, -%11, 1.0, %7 op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
            if (KIND == 4) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (KIND == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (KIND == 6) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
            if (KIND == 7) { REP8(asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %10\n v_cvt_f32_f16 %3, %11\n v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %9\n v_cvt_f32_f16 %6, %10\n v_cvt_f32_f16 %7, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        }
    } else if (MFMA) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {     // 8 MFMAs (2 accumulators) = 256 pipe cycles per iteration
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc1, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(d0 + d1 + d2 + d3);
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 123.456f) sink[threadIdx.x] = s;
}
template <int KIND, int MFMA> void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<KIND, MFMA>), dim3(256), dim3(512), 0, 0, d, sink, iters);
    hipLaunchKernelGGL((probe<KIND, MFMA>), dim3(256), dim3(512), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-14s partner %-5s: valu wave %6.2f cyc/instr   mfma wave %6.1f cyc/mfma\n", name, MFMA ? "mfma" : "idle",
           (double)h[0] / (iters * 64.0), MFMA ? (double)h[4] / (iters * 8.0) : 0.0);
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 4096);
#define BOTH(K, name) run<K, 0>(name, d, sink); run<K, 1>(name, d, sink);
    BOTH(0, "v_add_f32") BOTH(5, "v_fma_f32") BOTH(4, "v_max3_f32") BOTH(1, "v_exp_f32") BOTH(2, "v_cvt_pk_f16") BOTH(3, "v_fma_mix_f16") BOTH(7, "v_cvt_f32_f16") BOTH(6, "v_pk_add_f32")
    return 0;
}
