// Stand-alone check of the LDS-DMA staging variant of the ping-pong attention kernel (attention_f16x3.hip, template parameter DMA) against the
// register-staged variant: the same launches on the same inputs must give the same BITS (only the way a tile's bytes reach the ring differs), over
// shapes that exercise every branch of the staging schedule (1, 2, 3, 5 and 32 key tiles, a partial last tile, key masks, the key split), then the
// timing of both at the bench geometry.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/attn_dma_check.hip -o tools/probe/attn_dma_check.bin
#include "../../imp-release_amd/csrc/attention_f16x3.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
hipError_t imp_grant_dynamic_lds(const void* kernel, size_t bytes) { return bytes > 48 * 1024 ? hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipSuccess; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run_case(const char* name, int B, int nq, int nk, bool masked, bool split, int reps) {
    const int D = 256;
    const size_t nmax = (size_t)(nq > nk ? nq : nk);
    const size_t qkv = (size_t)B * nmax * 3 * D;
    std::vector<float> h(qkv);
    static unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.0f; }
    float *q0, *q1, *o[2][2], *lse[2][2];
    CK(hipMalloc(&q0, qkv * 4)); CK(hipMalloc(&q1, qkv * 4));
    CK(hipMemcpy(q0, h.data(), qkv * 4, hipMemcpyHostToDevice));
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.0f; }
    CK(hipMemcpy(q1, h.data(), qkv * 4, hipMemcpyHostToDevice));
    const size_t on = (size_t)B * nq * D, ln = (size_t)B * IMP_NUM_HEADS * nq;
    for (int v = 0; v < 2; ++v) for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&o[v][i], on * 4)); CK(hipMemset(o[v][i], 0xff, on * 4));
        CK(hipMalloc(&lse[v][i], ln * 4)); CK(hipMemset(lse[v][i], 0xff, ln * 4));
    }
    uint8_t* km = nullptr;
    if (masked) {
        std::vector<uint8_t> m((size_t)B * nk);
        for (auto& x : m) { s = s * 1664525u + 1013904223u; x = (s >> 13) % 3 != 0; }
        for (int b = 0; b < B; ++b) m[(size_t)b * nk + 5] = 1;
        CK(hipMalloc(&km, m.size())); CK(hipMemcpy(km, m.data(), m.size(), hipMemcpyHostToDevice));
    }
    AttnParams a;
    memset(&a, 0, sizeof a);
    a.nside = 2; a.ldq = a.ldk = 3 * D; a.ldo = D; a.dh = 64;
    float* qs[2] = {q0, q1};
    for (float* q : {q0, q1}) { CK(launch_attn_kv_planes(q, (long)B * nmax, 3 * D, D, 64, 0)); CK(launch_attn_kv_planes(q, (long)B * nmax, 3 * D, 2 * D, 64, 0)); }
    a.kv_planes = 1;
    if (split) {
        for (int i = 0; i < 2; ++i) { a.side[i].nq = nq; a.side[i].nk = nk; }
        const size_t fl = attention_f16x3_split_floats(a, B, 7), un = attention_f16x3_split_units(a, B);
        CK(hipMalloc(&a.split_ws, fl * 4)); CK(hipMalloc(&a.split_cnt, un * 4)); CK(hipMemset(a.split_cnt, 0, un * 4));
    }
    double us[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
        imp_attn_dma_override = v;
        for (int i = 0; i < 2; ++i) {
            AttnSide& g = a.side[i];
            g.q = qs[i]; g.k = qs[1 - i] + D; g.v = qs[1 - i] + 2 * D; g.out = o[v][i]; g.lse = lse[v][i]; g.kmask = km;
            g.sq_b = g.sk_b = (long)nmax * 3 * D; g.so_b = (long)nq * D; g.nq = nq; g.nk = nk;
        }
        CK(launch_attention_f16x3(a, B, 0));
        CK(hipDeviceSynchronize());
    }
    int bad = 0;
    for (int i = 0; i < 2; ++i) {
        std::vector<unsigned> x(on), y(on);
        CK(hipMemcpy(x.data(), o[0][i], on * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), o[1][i], on * 4, hipMemcpyDeviceToHost));
        size_t nd = 0, first = 0; int nanc = 0;
        for (size_t k = 0; k < on; ++k) { if (x[k] != y[k]) { if (!nd) first = k; ++nd; } if ((x[k] & 0x7f800000u) == 0x7f800000u) ++nanc; }
        std::vector<unsigned> lx(ln), ly(ln);
        CK(hipMemcpy(lx.data(), lse[0][i], ln * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ly.data(), lse[1][i], ln * 4, hipMemcpyDeviceToHost));
        size_t nl = 0;
        for (size_t k = 0; k < ln; ++k) nl += lx[k] != ly[k];
        if (nd || nl) {
            const size_t row = first / D, col = first % D;
            printf("  side %d: %zu of %zu outputs differ (first at pair %zu query %zu channel %zu: %08x vs %08x), %zu lse differ\n", i, nd, on, row / nq, row % nq, col, x[first], y[first], nl);
            bad = 1;
        }
        if (nanc) printf("  side %d: %d non-finite outputs in the register-staged result\n", i, nanc);
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep)
            for (int v = 0; v < 2; ++v) {
                imp_attn_dma_override = v;
                for (int i = 0; i < 2; ++i) { a.side[i].out = o[v][i]; a.side[i].lse = lse[v][i]; }
                CK(launch_attention_f16x3(a, B, 0));
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < reps; ++r) CK(launch_attention_f16x3(a, B, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us[v] = ms / reps * 1e3;
                printf("  pass %d %s: %.2f us / launch\n", rep, v ? "LDS-DMA        " : "register-staged", us[v]);
            }
    }
    printf("%-44s B=%d nq=%d nk=%d: %s\n", name, B, nq, nk, bad ? "DIFFERENT" : "bit-identical");
    for (int v = 0; v < 2; ++v) for (int i = 0; i < 2; ++i) { (void)hipFree(o[v][i]); (void)hipFree(lse[v][i]); }
    (void)hipFree(q0); (void)hipFree(q1); if (km) (void)hipFree(km);
    if (a.split_ws) { (void)hipFree(a.split_ws); (void)hipFree(a.split_cnt); }
    return bad;
}

int main() {
    int bad = 0;
    bad |= run_case("one key tile", 2, 256, 64, false, false, 0);
    bad |= run_case("two key tiles", 2, 256, 128, false, false, 0);
    bad |= run_case("three key tiles", 2, 256, 192, false, false, 0);
    bad |= run_case("five key tiles, partial last (300 keys)", 3, 520, 300, false, false, 0);
    bad |= run_case("partial last tile (1000 keys)", 2, 1000, 1000, false, false, 0);
    bad |= run_case("odd key count (2047)", 2, 2048, 2047, false, false, 0);
    bad |= run_case("key masks", 2, 1024, 1024, true, false, 0);
    bad |= run_case("key masks, partial last tile", 2, 700, 900, true, false, 0);
    bad |= run_case("key split (one pair, 1024)", 1, 1024, 1024, false, true, 0);
    bad |= run_case("key split, partial last tile (1500)", 1, 1500, 1500, false, true, 0);
    bad |= run_case("bench geometry", 4, 2048, 2048, false, false, 100);
    bad |= run_case("bench geometry, again (another input)", 4, 2048, 2048, false, false, 0);
    printf(bad ? "RESULT: the LDS-DMA variant DIFFERS from the register-staged kernel\n" : "RESULT: the LDS-DMA variant is bit-identical to the register-staged kernel on every case\n");
    return bad;
}
