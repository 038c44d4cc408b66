// Probe: cost of a hand-rolled device-wide (per-group) barrier + small all-to-all exchange on gfx950, 256 workgroups
// (one per CU), groups of G workgroups.  Every spin is bounded (a broken barrier exits with a flag, it cannot hang).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool group_barrier(unsigned* counter, unsigned target, int* fail) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: partials visible device-wide
        atomicAdd(counter, 1u);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 2000000) { *fail = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();                                   // acquire
    }
    __syncthreads();
    return true;
}

// mode 0: barrier only; mode 1: + write 8 KB partial, reduce a 32-column slice over G partial rows, write v slice, second barrier, read v
__global__ __launch_bounds__(512) void k(unsigned* counters, float* partials, float* v, int G, int iters, int mode, int* fail,
                                         unsigned long long* cycles) {
    const int grp = blockIdx.x / G, me = blockIdx.x % G;
    unsigned* cnt = counters + grp * 32;
    const int ld = 2052;
    float* mypart = partials + (size_t)blockIdx.x * ld;
    float* gv = v + (size_t)grp * ld;
    __shared__ float vs[2052];
    float acc = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) {
            for (int j = threadIdx.x; j < ld; j += 512) mypart[j] = acc + j + it;       // stand-in for the column partials
        }
        phase += G;
        group_barrier(cnt, phase, fail);
        if (mode == 1) {
            // reduce my slice of columns over the G partial rows (fixed order), write it
            const int cols_per = (ld + G - 1) / G;
            const int j = me * cols_per + (threadIdx.x % cols_per);
            if (threadIdx.x < cols_per && j < ld) {
                float s = 0.f;
                const float* pp = partials + (size_t)grp * G * ld + j;
                for (int w = 0; w < G; ++w) s += __builtin_nontemporal_load(pp + (size_t)w * ld);
                gv[j] = 1.f / (1.f + s * 1e-9f);
            }
            phase += G;
            group_barrier(cnt, phase, fail);
            for (int j2 = threadIdx.x; j2 < ld; j2 += 512) vs[j2] = __builtin_nontemporal_load(gv + j2);
            __syncthreads();
            acc = vs[(threadIdx.x * 7) % ld];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (acc == 12345.f) v[0] = acc;
}
int main() {
    unsigned* counters; float *partials, *v; int* fail; unsigned long long* cyc;
    CK(hipMalloc(&counters, 8 * 32 * 4)); CK(hipMalloc(&partials, 256 * 2052 * 4)); CK(hipMalloc(&v, 8 * 2052 * 4));
    CK(hipMalloc(&fail, 4)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {64, 256, 8}) {
            const int iters = 200;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(counters, 0, 8 * 32 * 4)); CK(hipMemset(fail, 0, 4));
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, counters, partials, v, G, iters, mode, fail, cyc);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int hf; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
                if (rep) printf("mode %d  group %3d: %.2f us per iteration%s\n", mode, G, ms * 1e3 / iters, hf ? "  (BARRIER TIMED OUT)" : "");
            }
        }
    return 0;
}
