// Stand-alone probe of the attention kernels (not part of the library): builds attention_f16x3.hip with PP_PROFILE and
// prints per-wave cycle counts of the ping-pong phases + launch timings.   hipcc --offload-arch=gfx950 -O3 -DPP_PROFILE
#include "../../imp-release_amd/csrc/attention_f16x3.hip"
#include <stdio.h>
// (the library keeps the granted sizes per device in gemm_f32.hip; the probe just grants)
hipError_t imp_grant_dynamic_lds(const void* kernel, size_t bytes) { return bytes > 48 * 1024 ? hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipSuccess; }
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, n = argc > 2 ? atoi(argv[2]) : 2048, D = 256;
    const int kvp = argc > 3 ? atoi(argv[3]) : 1;            // K / V as split-half images staged by plain copy (the product's format)
    const size_t qkv = (size_t)B * n * 3 * D;
    std::vector<float> h(qkv);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.0f; }
    float *q0, *q1, *o0, *o1;
    CK(hipMalloc(&q0, qkv * 4)); CK(hipMalloc(&q1, qkv * 4));
    CK(hipMalloc(&o0, (size_t)B * n * D * 4)); CK(hipMalloc(&o1, (size_t)B * n * D * 4));
    CK(hipMemcpy(q0, h.data(), qkv * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q1, h.data(), qkv * 4, hipMemcpyHostToDevice));
    AttnParams a;
    memset(&a, 0, sizeof a);
    a.nside = 2; a.ldq = a.ldk = 3 * D; a.ldo = D; a.dh = 64;
    float* qs[2] = {q0, q1}; float* os[2] = {o0, o1};
    for (int i = 0; i < 2; ++i) {
        AttnSide& g = a.side[i];
        g.q = qs[i]; g.k = qs[1 - i] + D; g.v = qs[1 - i] + 2 * D; g.out = os[i];
        g.sq_b = g.sk_b = (long)n * 3 * D; g.so_b = (long)n * D; g.nq = g.nk = n;
    }
    if (kvp) {
        for (float* q : {q0, q1}) { CK(launch_attn_kv_planes(q, (long)B * n, 3 * D, D, 64, 0)); CK(launch_attn_kv_planes(q, (long)B * n, 3 * D, 2 * D, 64, 0)); }
        a.kv_planes = 1;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        CK(launch_attention_f16x3(a, B, 0));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 100; ++r) CK(launch_attention_f16x3(a, B, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("B=%d n=%d: %.1f us/launch  %.1f TF\n", B, n, ms / 100 * 1e3, 2.0 * 4 * B * 2 * n * (double)n * 64 * 2 / (ms / 100) / 1e9);
    }
#ifdef PP_PROFILE
    unsigned long long prof[8][8];
    CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(pp_prof), sizeof prof));
    const int nt = (n + 63) / 64;
    printf("per tile (cycles of s_memtime): wave  X  barX  Y_softmax  Y_stage  barY  (loop) \n");
    for (int w = 0; w < 8; ++w)
        printf("  wave %d: X %6.0f  wait %6.0f  softmax %6.0f  stage %6.0f  wait %6.0f  other %6.0f\n", w, (double)prof[w][0] / nt,
               (double)prof[w][1] / nt, (double)prof[w][2] / nt, (double)prof[w][3] / nt, (double)prof[w][4] / nt, (double)prof[w][7] / nt);
    unsigned long long span[8][3];
    CK(hipMemcpyFromSymbol(span, HIP_SYMBOL(pp_span), sizeof span));
    printf("per wave of workgroup 0 (cycles): prologue (Q tile, first staged tiles, phase offset)  key loop  epilogue (normalise, transpose, store)\n");
    for (int w = 0; w < 8; ++w) printf("  wave %d: prologue %7llu  loop %8llu  epilogue %7llu\n", w, span[w][0], span[w][1], span[w][2]);
#endif
#ifdef PP_TIMELINE
    {
        unsigned long long tl[8][4][8]; unsigned hw[8];
        CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(pp_tl), sizeof tl));
        CK(hipMemcpyFromSymbol(hw, HIP_SYMBOL(pp_hwid), sizeof hw));
        unsigned long long base = ~0ull;
        for (int w = 0; w < 8; ++w) if (tl[w][0][0] < base) base = tl[w][0][0];
        printf("timeline of workgroup 0, key tiles %d..%d, cycles since the first stamp (X0 = matrix phase starts, X1 = its last MFMA issued, B1 = past the barrier,\n"
               " S = probabilities done, G = staging done, B2 = past the barrier); waves w and w+4 normally share a SIMD (HW_ID simd field printed)\n", PP_TL_T0, PP_TL_T0 + 3);
        for (int w = 0; w < 8; ++w) {
            printf("  wave %d simd %u cu %u:", w, (hw[w] >> 4) & 3, (hw[w] >> 8) & 15);
            for (int t = 0; t < 4; ++t) {
                printf("  |t%d", PP_TL_T0 + t);
                for (int i = 0; i < 6; ++i) printf(" %6lld", (long long)(tl[w][t][i] - base));
            }
            printf("\n");
        }
    }
#endif
    float chk = 0; std::vector<float> ho((size_t)B * n * D);
    CK(hipMemcpy(ho.data(), o0, ho.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ho.size(); i += 997) chk += ho[i];
    printf("checksum %.6f\n", chk);
    return 0;
}
