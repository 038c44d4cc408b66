// Stand-alone probe of the GEMM kernel on the three per-layer shapes of the benchmark (4 pairs x 2 images x 2048 tokens):
//   qkv  : [2048 x 256] . [768 x 256]^T            mlp0 : [2048 x (256 | 256)] . [512 x 512]^T (+ column statistics)
//   mlp3 : norm+relu([2048 x 512]) . [256 x 512]^T + residual
#include "../../imp-release_amd/csrc/gemm_f32.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static float* dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.f * scale; }
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    const int B = 4, n = 2048, D = 256;
    const int extra_flags = argc > 1 ? atoi(argv[1]) : 0;
    float* desc[2] = {dev_rand((size_t)B * n * D, 1, 1.f), dev_rand((size_t)B * n * D, 2, 1.f)};
    float* msg[2] = {dev_rand((size_t)B * n * D, 3, 1.f), dev_rand((size_t)B * n * D, 4, 1.f)};
    float* qkv[2] = {dev_rand((size_t)B * n * 3 * D, 5, 1.f), dev_rand((size_t)B * n * 3 * D, 6, 1.f)};
    float* hid[2] = {dev_rand((size_t)B * n * 2 * D, 7, 1.f), dev_rand((size_t)B * n * 2 * D, 8, 1.f)};
    float* out[2] = {dev_rand((size_t)B * n * D, 9, 1.f), dev_rand((size_t)B * n * D, 10, 1.f)};
    float* stats[2] = {dev_rand((size_t)B * 64 * 2 * D * 2, 11, 1.f), dev_rand((size_t)B * 64 * 2 * D * 2, 12, 1.f)};
    float* nstat[2] = {dev_rand((size_t)B * 2 * D * 2, 13, 1.f), dev_rand((size_t)B * 2 * D * 2, 14, 1.f)};
    float* Wqkv = dev_rand(768 * 256, 20, 0.06f), *W0 = dev_rand(512 * 512, 21, 0.04f), *W3 = dev_rand(256 * 512, 22, 0.04f);
    float* bias = dev_rand(768, 23, 0.1f);
    auto defaults = [&](int K) { GemmParams p; memset(&p, 0, sizeof p); p.K = K; p.ksplit = K; p.nside = 2; p.nsub = 1; p.prec = 1; p.norm_eps = 1e-3f; return p; };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, GemmParams& p, double bytes, double flops) {
        for (int rep = 0; rep < 3; ++rep) {
            launch_gemm_f32(p, B, 0);
            hipEventRecord(e0, 0);
            for (int r = 0; r < 50; ++r) launch_gemm_f32(p, B, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-6s %.1f us   %.2f TB/s algorithmic   %.0f TF(fp32-equivalent)\n", name, ms / 50 * 1e3, bytes / (ms / 50 * 1e-3) / 1e12, flops / (ms / 50 * 1e-3) / 1e12);
        }
#ifdef GEMM_PROFILE
        unsigned long long prof[4][8];
        hipMemcpyFromSymbol(prof, HIP_SYMBOL(gemm_prof), sizeof prof);
        const int nkt = p.K / 32;
        for (int w = 0; w < 4; ++w)
            printf("   wave %d per K-tile: store(wait+split+ds_write) %5.0f  barrier %5.0f  load-issue %5.0f  frags+mfma %5.0f  barrier %5.0f  loop %4.0f\n", w,
                   (double)prof[w][0] / nkt, (double)prof[w][1] / nkt, (double)prof[w][2] / nkt, (double)prof[w][3] / nkt, (double)prof[w][4] / nkt, (double)prof[w][5] / nkt);
#endif
    };
    const double M = 2.0 * B * n;
    {   // qkv
        GemmParams p = defaults(D);
        for (int s = 0; s < 2; ++s) { GemmSide& g = p.side[s]; g.A = desc[s]; g.W = Wqkv; g.M = n; g.N = 768; g.C = qkv[s]; g.sA_b = (long)n * D; g.sC_b = (long)n * 3 * D; }
        p.bias = bias; p.lda = D; p.ldw = D; p.ldc = 3 * D; p.flags |= extra_flags;
        timeit("qkv", p, M * (256 + 768) * 4, 2 * M * 256 * 768);
    }
    {   // mlp0
        GemmParams p = defaults(2 * D); p.ksplit = D;
        for (int s = 0; s < 2; ++s) { GemmSide& g = p.side[s]; g.A = desc[s]; g.A2 = msg[s]; g.W = W0; g.C = hid[s]; g.M = n; g.N = 2 * D; g.sA_b = (long)n * D; g.sC_b = (long)n * 2 * D; g.out_stats = stats[s]; }
        p.flags = GEMM_EPI_STATS | extra_flags; p.bias = bias; p.lda = D; p.lda2 = D; p.ldw = 2 * D; p.ldc = 2 * D;
        timeit("mlp0", p, M * (512 + 512) * 4, 2 * M * 512 * 512);
    }
    {   // mlp3
        GemmParams p = defaults(2 * D); p.flags = GEMM_PRO_NORM | extra_flags; p.act = 0;
        for (int s = 0; s < 2; ++s) { GemmSide& g = p.side[s]; g.A = hid[s]; g.W = W3; g.C = out[s]; g.R = desc[s]; g.M = n; g.N = D; g.sA_b = (long)n * 2 * D; g.sC_b = (long)n * D; g.sR_b = (long)n * D; g.in_stats = nstat[s]; }
        p.bias = bias; p.lda = 2 * D; p.ldw = 2 * D; p.ldc = D; p.ldr = D;
        timeit("mlp3", p, M * (512 + 256 + 256) * 4, 2 * M * 512 * 256);
    }
    return 0;
}
