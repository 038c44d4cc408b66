// Multi-head attention core (nets/layers.py:121-131), f16x3 arithmetic, ONE WAVE PER SIMD (round 6).
//
// attention_f16x3.hip's ping-pong kernel pairs two waves on every SIMD - one in its matrix phase, one in its vector phase - and was measured to
// its wall in round 5 (profiles/r05: every stall removed from one wave is taken over by its partner, both share the SIMD's VALU issue, and each of
// the 8 waves reads every K / V tile from LDS).  This kernel is the structure after that wall, for the launches that carry the benchmark
// (split-half K / V images, head width 64, no key mask, no key split):
//   * a workgroup is still 256 queries on one CU, but 4 waves, one per SIMD, 64 queries each: TWO 32-query blocks per wave, so every K / V
//     fragment fetched from LDS feeds the MFMAs of both blocks - half the LDS fragment traffic per query (262 KB per tile and CU before);
//   * the whole 512-entry register file belongs to the wave (launch bounds 256 x 1): the logits of tile t + 1 accumulate in one buffer while the
//     probabilities of tile t are still being finished in the other, so the softmax never waits for the matrix pipe or the other way round;
//   * the vector work is dealt into the gaps of the wave's OWN MFMA stream, at most a handful of issues per gap:
//       A(t + 1):  S(t + 1)^T = K(t + 1) . Q^T   (48 MFMAs)  ||  hi / lo split of the probabilities of tile t, K fragment reads, global loads of tile t + 3
//       B(t):      O^T += V(t - 1)^T . P(t - 1)^T (48 MFMAs)  ||  exp2 + row sums of tile t, V fragment reads, LDS stores of tile t + 2
//     one workgroup barrier per key tile.
// The arithmetic of a 32-query block is, instruction for instruction per value, the ping-pong kernel's: the same fragment images, the same MFMA
// order per accumulator (lo.hi, hi.lo, hi.hi per k-step), the same lazily updated softmax reference (the slow path of a tile whose row sums
// outgrow it is taken per 32-query block, on logits recomputed from the tile still in the ring), the same sequential row sums, the same division:
// results are BIT-IDENTICAL to attention_f16x3.hip's, which remains the kernel of every other launch (masks, key splits, head width 32, fp32 K / V)
// - a pair's output does not depend on which of the two kernels a launch took (tests/test_gpu_parity.py
// test_one_wave_per_simd_attention_kernel_is_bit_identical_to_the_ping_pong_kernel).
#include "imp_kernels.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

namespace {

constexpr int KT = 64, DH = 64, KS = 4, DT = 2, NT = 256;
constexpr int KROW = DH + 4;                 // K row: 32 floats of hi halves, 32 of lo halves, 4 pad (the ring images of attention_f16x3.hip)
constexpr int VROW = DH + 16;                // V row: same split, padded to 320 B (conflict-free transpose reads)
constexpr int LK = KT * DH / 4 / NT;         // float4 of K (and of V) per thread and tile: 4
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SL2E = 0.125f * LOG2E;       // 1 / sqrt(DH) * log2(e)
constexpr float P_SUM_LIMIT = 16384.f;       // per-lane partial row sum that forces a reference update

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = lin % 8, idx = lin / 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
// The MFMAs are written as inline asm with REGISTER-CLASS constraints: one wave per SIMD owns 256 architectural + 256 accumulator registers, and only
// the former can be VALU operands.  Left to the allocator (builtin MFMAs) the logits landed in accumulator registers and every exp2 / split paid a
// v_accvgpr_read / _write (1 900 such moves in the loop body of the first build).  Here: the logits / probabilities S (VALU reads and writes them) and the
// P fragments (VALU writes them) are VGPRs; O (touched by the VALU only on the slow path and in the epilogue), Q and the K / V fragments (ds_read can
// target accumulator registers directly) are AGPRs.  C and D of an MFMA share one class bit (ACC_CD), so S chains start from a VGPR pre-set to
// -m_ref (no separate constant C operand).  Hazards: every consumer of an MFMA result sits >= 4 independent MFMAs (128 cycles) behind it or behind
// an explicit drain (w4_drain).
__device__ __forceinline__ void mfma_s(f32x16& acc, const f16x8& kfrag, const f16x8& q) {          // S += K . Q^T : acc VGPR, operands AGPR
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(kfrag), "a"(q));
}
__device__ __forceinline__ void mfma_o(f32x16& acc, const f16x8& vfrag, const f16x8& pf) {         // O += V^T . P^T : acc AGPR, V fragment AGPR, P fragment VGPR
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(vfrag), "v"(pf));
}
__device__ __forceinline__ void w4_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }     // the last MFMAs of a block have written back

__global__ __launch_bounds__(NT, 1) void attn_f16x3_w4_kernel(const AttnParams p, int qtiles, int total_blocks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                           // [4][KT][KROW]
    float* Vs = Ks + 4 * KT * KROW;             // [4][KT][VROW]   key-major: the PV operand is read transposed
    float* Bs = Vs + 4 * KT * VROW;             // [4][KT]         0 / -inf per key (a partial last tile)

    const bool clk_on = p.clk_probe != nullptr && blockIdx.x == 0;      // workgroup-uniform
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = xcd_remap(blockIdx.x, total_blocks);
    const int qt = id % qtiles; id /= qtiles;
    const int h = id % IMP_NUM_HEADS; id /= IMP_NUM_HEADS;
    const int sidx = id % p.nside;
    const int b = id / p.nside;
    const AttnSide& S = p.side[sidx];
    const int nq = imp_count(p.rc, S.qimg, b, S.nq), nk = imp_count(p.rc, S.kimg, b, S.nk);      // ragged batches: this pair's own counts
    const int q0 = qt * 256;
    if (q0 >= nq || nk <= 0) return;
    const int nt = (nk + KT - 1) / KT;
    const float* Qg = S.q + b * S.sq_b + h * DH;
    const float* Kg = S.k + b * S.sk_b + h * DH;
    const float* Vg = S.v + b * S.sk_b + h * DH;

    // ---- Q rows of both 32-query blocks (requested first; split below, behind the first staged tiles' requests)
    f32x4 qraw[2][KS][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + wave * 64 + qb * 32 + l31;
        const float* src = Qg + (long)(qrow < nq ? qrow : nq - 1) * p.ldq + 8 * half;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qraw[qb][s][0] = *reinterpret_cast<const f32x4*>(src + 16 * s);
            qraw[qb][s][1] = *reinterpret_cast<const f32x4*>(src + 16 * s + 4);
        }
    }

    // ---- staging: the rows of K / V already ARE the ring's [hi | lo] images (AttnParams::kv_planes): thread = 4 sixteen-byte chunks of K and 4 of V per tile
    const unsigned kv_bytes = (unsigned)(((long)(nk - 1) * p.ldk + DH) * 4);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, kv_bytes, 0x00020000);
    const int row_bytes = p.ldk * 4;
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;          // chunk j of this thread: row srow + 16 j, floats sc4 .. sc4 + 3
    const int koff = srow * row_bytes + (tid & 15) * 16;
    f32x4 rk[LK], rv[LK];
    unsigned char rb = 1;
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        const int soff = t * KT * row_bytes;
#pragma unroll
        for (int j = 0; j < LK; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsK, koff + 16 * j * row_bytes, soff, 0);
            rk[j] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
#pragma unroll
        for (int j = 0; j < LK; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsV, koff + 16 * j * row_bytes, soff, 0);
            rv[j] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
        if (tid < KT) rb = (t * KT + tid) < nk;
    };
    auto store_tile = [&](int slot) __attribute__((always_inline)) {
        float* ks = Ks + slot * KT * KROW + srow * KROW + sc4;
        float* vs = Vs + slot * KT * VROW + srow * VROW + sc4;
#pragma unroll
        for (int j = 0; j < LK; ++j) {
            *reinterpret_cast<f32x4*>(ks + 16 * j * KROW) = rk[j];
            *reinterpret_cast<f32x4*>(vs + 16 * j * VROW) = rv[j];
        }
        if (tid < KT) Bs[slot * KT + tid] = rb ? 0.f : -INFINITY;
    };

    // ---- state of the two 32-query blocks of this wave (block qb = wave 2 * wave + qb of the ping-pong kernel)
    f32x16 oacc[2][DT];                    // O^T accumulators
    f32x16 sA[2][2], sB[2][2];             // [qb][key block of 32]: logits of the tile being accumulated / probabilities of the tile being finished
    f16x8 qh[2][KS], ql[2][KS];
    f16x8 ph[2][2][2], pl[2][2][2];        // [qb][jb][s2]: B operands of O^T += V^T . P^T
    float m_ref[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
    // a logits buffer starts a tile at -m_ref (the accumulators then deliver log2-domain logits relative to the reference: P = exp2(acc))
    auto init_s = [&](f32x16 (&sb)[2][2], int qb) __attribute__((always_inline)) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[qb][jb][r] = -m_ref[qb];
    };

    load_tile(0);
    // (Q split: the ping-pong kernel's, per 32-query block)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 a = qraw[qb][s][0], c = qraw[qb][s][1];
            const float x[8] = {a[0] * SL2E, a[1] * SL2E, a[2] * SL2E, a[3] * SL2E, c[0] * SL2E, c[1] * SL2E, c[2] * SL2E, c[3] * SL2E};
            u32x4 hh, ll;
#pragma unroll
            for (int i = 0; i < 4; ++i) { unsigned u, v; imp_split2(x[2 * i], x[2 * i + 1], u, v); hh[i] = u; ll[i] = v; }
            qh[qb][s] = __builtin_bit_cast(f16x8, hh);
            ql[qb][s] = __builtin_bit_cast(f16x8, ll);
            asm volatile("" : "+a"(qh[qb][s]), "+a"(ql[qb][s]));        // pinned to accumulator-register tuples once (else every use copies the fragment into a fresh tuple)
        }
    store_tile(0);
    if (nt > 1) { load_tile(1); store_tile(1); }
    if (nt > 2) load_tile(2);                         // in flight until iteration 0 stores it
    __syncthreads();

    // ---- fragment reads (addresses of attention_f16x3.hip: every 16-lane group hands in a [4 keys][16 channels] block for the transpose read)
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int vlane = ((4 * half + ((lane & 15) >> 2)) * VROW) * 4 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto read_v1 = [&](int slot, int g, int i, f16x8& f) __attribute__((always_inline)) {       // i: 0, 1 = vh[d], 2, 3 = vl[d] of k-step g
        const char* a = reinterpret_cast<const char*>(Vs + slot * KT * VROW) + vlane + (16 * g) * (VROW * 4) + (i % DT) * 64 + (i / DT) * (DH * 2);
        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
        const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 8 * VROW * 4));
        f = __builtin_bit_cast(f16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto read_k1 = [&](int slot, int s, int i, f16x8& f) __attribute__((always_inline)) {       // i: 0, 1 = kh[jb], 2, 3 = kl[jb] of k-step s
        const float* ks = Ks + slot * KT * KROW + l31 * KROW + 4 * half + 8 * s;
        f = *reinterpret_cast<const f16x8*>(ks + (i & 1) * 32 * KROW + (i >> 1) * (DH / 2));
    };
    f16x8 fr[2][4];                                   // fragment double buffer

    // ---- S(slot)^T += K . Q^T for the blocks in `mask` (bit qb) into s[qb][jb], which the caller has pre-set to -m_ref; `fill(gap)` is dealt one call per MFMA gap (gap = 12 s + i)
    auto qk_block = [&](int slot, f32x16 (&s)[2][2], int mask, auto&& fill) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) read_k1(slot, 0, i, fr[0][i]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 (&f)[4] = fr[ks & 1];
            f16x8 (&n)[4] = fr[(ks + 1) & 1];
            auto rd = [&](int i) __attribute__((always_inline)) { if (ks + 1 < KS) read_k1(slot, ks + 1, i, n[i]); };
            int gap = 12 * ks;
            W4_SB();
            // per accumulator the ping-pong kernel's order: kl . qh, kh . ql, kh . qh
#pragma unroll
            for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        if (mask & (1 << qb)) {
                            mfma_s(s[qb][jb], prod == 0 ? f[2 + jb] : f[jb], prod == 1 ? ql[qb][ks] : qh[qb][ks]);
                            W4_SB();
                        }
                        const int g4 = prod * 4 + qb * 2 + jb;
                        if (g4 < 4) { rd((g4 + 2) & 3); W4_SB(); }          // read order 2, 3, 0, 1: the order in which the MFMAs of a step first touch them
                        fill(gap + g4);
                        W4_SB();
                    }
        }
    };
    // ---- O^T += V(slot)^T . P^T for both blocks; `fill(gap)` as above (gap = 12 g + i)
    auto pv_block = [&](int slot, auto&& fill) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) read_v1(slot, 0, i, fr[0][i]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int jb = g >> 1, s2 = g & 1;
            f16x8 (&f)[4] = fr[g & 1];
            f16x8 (&n)[4] = fr[(g + 1) & 1];
            auto rd = [&](int i) __attribute__((always_inline)) { if (g < 3) read_v1(slot, g + 1, i, n[i]); };
            W4_SB();
            // per accumulator: vl . ph, vh . pl, vh . ph
#pragma unroll
            for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        mfma_o(oacc[qb][d], prod == 0 ? f[DT + d] : f[d], prod == 1 ? pl[qb][jb][s2] : ph[qb][jb][s2]);
                        W4_SB();
                        const int g4 = prod * 4 + qb * 2 + d;
                        if (g4 < 4) { rd((g4 + 2) & 3); W4_SB(); }
                        fill(12 * g + g4);
                        W4_SB();
                    }
        }
    };
    auto nofill = [](int) {};

    // ---- vector work in units the gaps can take
    // exp2 of two logits of block qb in place + their addition to the block's sequential row sum (order of the ping-pong kernel: jb, s2, e ascending)
    float lsum[2];
    auto exp_pair = [&](f32x16 (&s)[2][2], int qb, int i) __attribute__((always_inline)) {       // i = 0 .. 15: values 2 i, 2 i + 1 of the block's 32
        const int jb = i >> 3, r = 2 * (i & 7);
        const float a = fast_exp2(s[qb][jb][r]), c = fast_exp2(s[qb][jb][r + 1]);
        s[qb][jb][r] = a; s[qb][jb][r + 1] = c;
        lsum[qb] += a;
        lsum[qb] += c;
    };
    // hi / lo split of two finished probabilities into the P fragments
    u32x4 sph[2][2][2], spl[2][2][2];
    auto split_pair = [&](f32x16 (&s)[2][2], int qb, int i) __attribute__((always_inline)) {
        const int jb = i >> 3, s2 = (i >> 2) & 1, k = i & 3, r = 8 * s2 + 2 * k;
        unsigned hi, lo;
        imp_split2(s[qb][jb][r], s[qb][jb][r + 1], hi, lo);
        s[qb][jb][r] = -m_ref[qb]; s[qb][jb][r + 1] = -m_ref[qb];            // this buffer's next tile (t + 2) accumulates from the reference
        sph[qb][jb][s2][k] = hi; spl[qb][jb][s2][k] = lo;
        if (k == 3) {
            ph[qb][jb][s2] = __builtin_bit_cast(f16x8, sph[qb][jb][s2]);
            pl[qb][jb][s2] = __builtin_bit_cast(f16x8, spl[qb][jb][s2]);
        }
    };
    // a partial last tile: keys past nk get -inf (the ping-pong kernel adds the tile's bias vector whenever the tile is partial)
    auto add_bias = [&](f32x16 (&s)[2][2], int slot) __attribute__((always_inline)) {
        const float* bs = Bs + slot * KT;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + jb * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[qb][jb][4 * g + e] += bias[e];
            }
    };
    // the slow path of one block on INTACT logits (relative to the current reference): exact tile maximum, reference raised, O and l rescaled,
    // probabilities relative to the new reference in place; returns whether the block still waits for its first unmasked key
    auto slow_path = [&](f32x16 (&s)[2][2], f32x16 (&other)[2][2], int qb) __attribute__((always_inline)) -> bool {
        float tmax = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qb][jb][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float lq = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float delta = (tmax == -INFINITY) ? 0.f : (lq > 0.f ? fmaxf(tmax, 0.f) : tmax);
        const float alpha = lq > 0.f ? fast_exp2(-delta) : 0.f;
        m_ref[qb] += delta;
        l_run[qb] *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
        float ls = 0.f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fast_exp2(s[qb][jb][r] - delta);
                s[qb][jb][r] = v;
                ls += v;
            }
        lsum[qb] = ls;
        init_s(other, qb);                                    // the buffer of the next tile was pre-set to the OLD reference
        return __any(tmax == -INFINITY && !(lq > 0.f)) != 0;
    };

    // ---- tile 0: plain (every block takes the slow path on its first tile)
    const bool partial_last = nt * KT > nk;
    bool need_slow[2];
    init_s(sA, 0); init_s(sA, 1);
    qk_block(0, sA, 3, nofill);
    w4_drain();
    if (nt == 1 && partial_last) add_bias(sA, 0);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) need_slow[qb] = slow_path(sA, sB, qb);

    // one iteration: `cur` holds the probabilities of tile t (finished but for the split), `nxt` receives the logits of tile t + 1
    auto iteration = [&](auto first, int t, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2]) __attribute__((always_inline)) {
        if constexpr (!decltype(first)::value) {
            // ======== B(t): O^T += V(t - 1)^T . P(t - 1)^T  ||  exp2 + row sums of tile t ========
            if (t == nt - 1 && partial_last) add_bias(cur, t & 3);
            const bool fast0 = !need_slow[0], fast1 = !need_slow[1];               // wave-uniform (false only while a block has seen no key at all: never past tile 0 without masks)
            lsum[0] = lsum[1] = 0.f;
            if (fast0 && fast1) {
                pv_block((t - 1) & 3, [&](int gap) __attribute__((always_inline)) {
                    // 48 gaps, 32 exp pairs: gaps 4 .. 11 of every k-step carry one pair each (the first four carry the fragment reads)
                    const int g = gap / 12, i = gap % 12;
                    if (i >= 4) { const int u = 8 * g + (i - 4); exp_pair(cur, u >> 4, u & 15); }
                });
            } else {
                pv_block((t - 1) & 3, nofill);
                if (fast0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) exp_pair(cur, 0, i);
                }
                if (fast1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) exp_pair(cur, 1, i);
                }
            }
            // a block whose row sums outgrew the reference (or that has no reference yet) redoes the tile on the slow path: its logits are
            // recomputed from the tile still in the ring (deterministic: the same bits), then treated like the ping-pong kernel treats them
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const bool fast = qb == 0 ? fast0 : fast1;
                bool redo = !fast;
                if (fast) redo = __any(!(lsum[qb] < P_SUM_LIMIT)) != 0;             // (also catches inf / nan)
#ifdef W4_PROBE_HOT
                redo = false;                                                      // (ISA inspection only: the loop without its cold paths)
#endif
                if (redo) {
                    if (fast) {
                        init_s(cur, qb);
                        qk_block(t & 3, cur, 1 << qb, nofill);
                        w4_drain();
                        if (t == nt - 1 && partial_last) {
                            const float* bs = Bs + (t & 3) * KT;
#pragma unroll
                            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + jb * 32 + 8 * g + 4 * half);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) cur[qb][jb][4 * g + e] += bias[e];
                                }
                        }
                    }
                    need_slow[qb] = slow_path(cur, nxt, qb);
                }
            }
        }
        l_run[0] += lsum[0];
        l_run[1] += lsum[1];
        // ======== staging: tile t + 2 into the ring, tile t + 3 requested ========
        if (t + 2 < nt) store_tile((t + 2) & 3);
        if (t + 3 < nt) load_tile(t + 3);
        // ======== A(t + 1): S(t + 1)^T = K(t + 1) . Q^T  ||  hi / lo split of tile t's probabilities ========
        if (t + 1 < nt) {
            qk_block((t + 1) & 3, nxt, 3, [&](int gap) __attribute__((always_inline)) {
                // 48 gaps, 32 split pairs: gaps 4 .. 11 of every k-step
                const int ks = gap / 12, i = gap % 12;
                if (i >= 4) { const int u = 8 * ks + (i - 4); split_pair(cur, u >> 4, u & 15); }
            });
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) split_pair(cur, u >> 4, u & 15);
        }
        __syncthreads();
    };
    // (tile 0's row sums come out of its slow path above: iteration 0 adds them like any other's)
    iteration(std::true_type{}, 0, sA, sB);           // (peeled: no P.V yet - the loop body then has no conditional around its matrix blocks)
    for (int t = 1; t < nt; t += 2) {
        iteration(std::false_type{}, t, sB, sA);
        if (t + 1 < nt) iteration(std::false_type{}, t + 1, sA, sB);
    }
    pv_block((nt - 1) & 3, nofill);
    w4_drain();
    __syncthreads();                                  // everyone is done with the ring: reuse it for the transposition

    // ---- epilogue per 32-query block: O / l (one IEEE division + Markstein's correction per element: bit-identical to the division), rows through LDS
    constexpr int LDP = DH + 4, LPR = DH / 4, RPI = 64 / LPR;
    const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
    float* Og = S.out + b * S.so_b + h * DH;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int w8 = 2 * wave + qb;                 // the wave of the ping-pong kernel that owns these 32 queries
        float* otp = smem + w8 * 32 * LDP;
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv_l = 1.0f / l_tot;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) otp[l31 * LDP + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = imp_div_by(oacc[qb][d][r], l_tot, inv_l);
        if (S.lse && half == 0) {
            const int qrow = q0 + w8 * 32 + l31;
            if (qrow < nq) S.lse[((long)b * IMP_NUM_HEADS + h) * S.nq + qrow] = m_ref[qb] * (1.0f / LOG2E) + logf(l_tot);
        }
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int w8 = 2 * wave + qb;
        const float* otp = smem + w8 * 32 * LDP;
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j) {
            const int qi = j * RPI + prow;
            const int qrow = q0 + w8 * 32 + qi;
            const f32x4 v = *reinterpret_cast<const f32x4*>(otp + qi * LDP + pc4);
            if (qrow < nq) *reinterpret_cast<f32x4*>(Og + (long)qrow * p.ldo + pc4) = v;
        }
    }
    if (clk_on && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        p.clk_probe[0] += __builtin_readcyclecounter() - clk_c0;
        p.clk_probe[1] += __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}

}  // namespace

// can this launch take the one-wave-per-SIMD kernel?  (split-half K / V images at head width 64, no key mask, no key split, 16-byte aligned rows)
bool attention_f16x3_w4_ok(const AttnParams& p, int nsplit) {
    if (p.dh != 64 || !p.kv_planes || nsplit > 1 || (p.ldk & 3) || (p.ldq & 3) || (p.ldo & 3)) return false;
    for (int s = 0; s < p.nside; ++s) {
        const AttnSide& g = p.side[s];
        if (g.kmask) return false;
        if ((reinterpret_cast<size_t>(g.k) | reinterpret_cast<size_t>(g.v) | reinterpret_cast<size_t>(g.q) | reinterpret_cast<size_t>(g.out) |
             (size_t)(g.sk_b * 4) | (size_t)(g.sq_b * 4) | (size_t)(g.so_b * 4)) & 15) return false;
    }
    return true;
}

hipError_t launch_attention_f16x3_w4(const AttnParams& p, int batch, int maxq, hipStream_t stream) {
    const int qtiles = (maxq + 255) / 256;
    const int total = qtiles * IMP_NUM_HEADS * p.nside * batch;
    const size_t lds = (size_t)(4 * KT * KROW + 4 * KT * VROW + 4 * KT) * sizeof(float);
    if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_w4_kernel, lds)) return e;
    hipLaunchKernelGGL(attn_f16x3_w4_kernel, dim3(total), dim3(NT), lds, stream, p, qtiles, total);
    return hipGetLastError();
}
