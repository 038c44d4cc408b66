// Probe 2 (round 2): a fence-free group barrier on gfx950.  The round-1 probe paid ~8 us per barrier because
// __threadfence() at agent scope writes back the XCD's L2.  Here the exchanged data moves with agent-scope relaxed atomic
// stores / loads (write-through / L2-bypassing sc1 accesses), ordered by s_waitcnt vmcnt(0) + the workgroup barrier, and the
// counter is a relaxed agent-scope atomic.  Every spin is bounded.  The exchange is CHECKED (the reduced value is known).
//   mode 0: barrier only     mode 1: the Sinkhorn-shaped exchange (8 KB partial per WG -> slice reduce -> v slice -> all read v)
//   placement 0: group = G consecutive blocks (spread over the 8 XCDs)   1: group = blocks with the same blockIdx % 8 (one XCD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void group_barrier(unsigned* counter, unsigned target, int* fail) {
    __builtin_amdgcn_s_waitcnt(0);          // this thread's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 4000000) { *fail = 1; break; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k(unsigned* counters, float* partials, float* v, int G, int iters, int mode, int placement,
                                         int* fail, int* wrong) {
    int grp, me;
    if (placement == 0) { grp = blockIdx.x / G; me = blockIdx.x % G; }
    else { grp = blockIdx.x % 8; me = blockIdx.x / 8; if (me >= G) return; }
    unsigned* cnt = counters + grp * 64;
    const int ld = 2048;
    float* mypart = partials + ((size_t)grp * G + me) * ld;
    float* gv = v + (size_t)grp * ld;
    __shared__ float vs[2048];
    unsigned phase = 0;
    float vloc = 1.f;
    for (int it = 0; it < iters; ++it) {
        if (mode == 1)
            for (int j = threadIdx.x; j < ld; j += 512) st_agent(mypart + j, (float)(me + 1) * vloc + (float)(j & 7));
        phase += G;
        group_barrier(cnt, phase, fail);
        if (mode == 1) {
            const int cols_per = ld / G;
            if ((int)threadIdx.x < cols_per) {
                const int j = me * cols_per + threadIdx.x;
                float s = 0.f;
                const float* pp = partials + (size_t)grp * G * ld + j;
                for (int w = 0; w < G; ++w) s += ld_agent(pp + (size_t)w * ld);
                const float expect = vloc * (float)(G * (G + 1) / 2) + (float)(j & 7) * G;
                if (s != expect) atomicAdd(wrong, 1);
                st_agent(gv + j, (float)(1 + (it & 3)));
            }
            phase += G;
            group_barrier(cnt, phase, fail);
            for (int j2 = threadIdx.x; j2 < ld; j2 += 512) vs[j2] = ld_agent(gv + j2);
            __syncthreads();
            vloc = vs[(threadIdx.x * 7) % ld];
            if (vloc != (float)(1 + (it & 3))) atomicAdd(wrong, 1);
            __syncthreads();
        }
    }
}
int main() {
    unsigned* counters; float *partials, *v; int *fail, *wrong;
    CK(hipMalloc(&counters, 8 * 64 * 4)); CK(hipMalloc(&partials, 256 * 2048 * 4)); CK(hipMalloc(&v, 8 * 2048 * 4));
    CK(hipMalloc(&fail, 4)); CK(hipMalloc(&wrong, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int placement = 0; placement < 2; ++placement)
        for (int mode = 0; mode < 2; ++mode)
            for (int G : {64, 32, 16}) {
                if (placement == 1 && G > 32) continue;
                const int iters = 200;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(counters, 0, 8 * 64 * 4)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(wrong, 0, 4));
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, counters, partials, v, G, iters, mode, placement, fail, wrong);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    int hf, hw; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hw, wrong, 4, hipMemcpyDeviceToHost));
                    if (rep) printf("placement %d mode %d group %3d: %.2f us per iteration%s wrong=%d\n", placement, mode, G,
                                    ms * 1e3 / iters, hf ? "  (BARRIER TIMED OUT)" : "", hw);
                }
            }
    return 0;
}
