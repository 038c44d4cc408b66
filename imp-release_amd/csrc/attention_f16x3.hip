// Multi-head attention core (nets/layers.py:121-131), split-precision variant of attention_f32.hip.
//
// Same decomposition (one wave = 32 queries, "swapped" S^T = K.Q^T so the query is the lane, 64-key tiles
// double-buffered in LDS, online softmax in registers) but every fp32 operand x is carried as two halves
// hi = f16(x), lo = f16(x - hi) (22 significant bits) and every fp32 product becomes three f16 MFMAs
// (lo.hi + hi.lo + hi.hi, fp32 accumulate, v_mfma_f32_32x32x16_f16): fp32-level results at 3/16 of the
// fp32-MFMA pipe time.  Validated end to end against the reference fixtures (tests/) - bf16 x3 is NOT enough.
//
// LDS images (per 64-key tile; both 272-byte rows = 68 floats, so 16 consecutive rows hit 16 distinct 16B slots):
//   K  [key][ hi: DH halves | lo: DH halves | pad ]      A operand of S^T: lane (key, half h) reads 8 halves
//   V^T[d  ][ hi: 64 key halves | lo: 64 | pad ]         A operand of O^T += V^T.P^T
// V is transposed while staging (thread = one channel d x 16 consecutive keys, coalesced dword loads) and the
// key positions inside each 32-key block have bits 2 and 3 swapped: the MFMA C layout leaves lane-half h with
// keys {4h + 8g + e}, so after the swap the 8 keys a lane needs for one 16-deep k-step are contiguous = ONE
// ds_read_b128 per operand, and the probabilities are used straight from the accumulator registers.
#include "imp_kernels.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KT = 64;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = lin % 8, idx = lin / 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
// 8 consecutive operand values -> one MFMA fragment of hi halves and one of lo halves
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned a, b; imp_split2(x[2 * i], x[2 * i + 1], a, b); h[i] = a; l[i] = b; }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}
__device__ __forceinline__ void split4(const f32x4 x, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { unsigned a, b; imp_split2(x[2 * i], x[2 * i + 1], a, b); hi[i] = a; lo[i] = b; }
}

// a wave's 32 output rows (staged in LDS as ot[row][DH]) -> memory as fp32 rows
template <int DH>
__device__ __forceinline__ void store_attention_rows(const AttnParams& p, const AttnSide& S, int b, int h, int qbase, int nq,
                                                     const float* ot, int ldot, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int RPP = 64 / DH;                 // rows per pass: DH = 64 -> 1 (lane = channel), DH = 32 -> 2
    float* Og = S.out + b * S.so_b + h * DH;
#pragma unroll 4
    for (int i = 0; i < 32 / RPP; ++i) {
        const int qi = RPP == 1 ? i : 2 * i + half, ch = RPP == 1 ? lane : l31;
        const int qrow = qbase + qi;
        if (qrow < nq) Og[(long)qrow * p.ldo + ch] = ot[qi * ldot + ch];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Ping-pong variant (DH = 64, 8 waves = 256 queries per workgroup, one workgroup per CU).
//
// When the two waves that share a SIMD run in lock step (one barrier per key tile: the kernel of rounds 1-5 for small launches), the matrix
// pipe idles while both do their softmax / split VALU work and the VALU idles while both issue MFMAs - and with
// the split-precision scheme the vector work is the larger half (measured on gfx950: a VALU instruction of a wave
// whose SIMD partner streams MFMAs costs ~6 cycles, v_exp_f32 / v_fma_mix ~10; 48 MFMAs = 1536 pipe cycles).
// Here every key tile is cut into a matrix phase X and a vector phase Y,
//     X(t): O^T += V(t-1)^T . P(t-1)^T   then   S(t)^T = K(t) . Q^T      (48 MFMAs, LDS fragment reads; it also
//           issues the global loads of tile t+3 - VMEM issue is slow and the matrix phase has the issue slots)
//     Y(t): softmax numerators of S(t) -> P(t) as hi/lo halves in registers, conversion of tile t+2 into LDS,
// with a workgroup barrier after every phase, and waves 4-7 pass ONE extra barrier before their loop: they run
// exactly one phase behind waves 0-3 for the whole kernel, so on every SIMD one wave is in X while its partner is
// in Y.  Barrier-phase index of a phase: waves 0-3 X(t) = 2t, Y(t) = 2t+1; waves 4-7 X(t) = 2t+1, Y(t) = 2t+2.
// LDS ring of 4 tiles (slot = t & 3; K rows [hi | lo] halves, V rows the same, key-major): tile t is written in phases 2t-3 / 2t-2 (each group stages the half of the
// tile its threads own), K(t) is read in phases 2t / 2t+1, V(t) in 2t+2 / 2t+3, and the slot is next written for
// tile t+4 in phase 2t+5 - every write is separated from every read of the previous occupant by a barrier.
// Two staged tiles are in flight in registers: a tile is loaded three phases before it is converted.
//
// Softmax with a lazily updated reference (removes the per-tile max, scale and rescale work from the common path):
// Q is pre-multiplied by scale * log2(e), the S accumulators are INITIALISED to -m_ref (the query's reference
// exponent, per lane), so the MFMAs deliver log2-domain logits relative to m_ref and P = exp2(acc) directly.
// m_ref is only raised when a tile's probabilities get large: the row sums are computed anyway, and a wave
// whose partial sums stay below 2^14 has every P < 2^14 (no f16 overflow, hi/lo relative precision unchanged);
// otherwise - and for the first tile(s), until every query of the wave has seen an unmasked key - the tile takes the
// slow path: exact tile maximum, m_ref += delta, O and l rescaled by 2^-delta, P recomputed from the same registers.
// ------------------------------------------------------------------------------------------------------------------
#define PP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// (round 5 measured a variant whose ring is filled by LDS-DMA - global_load_lds_dwordx4 - instead of through registers: bit-identical, 21 VGPRs fewer, -2.6 % cycles
// per phase, EQUAL in time, the launch being power-limited; removed from the product in round 6: tools/probe/attn_dma/, profiles/r05/attn_lds_dma_*.log)
template <int DH, bool MASKED>
__global__ __launch_bounds__(512, 2) void attn_f16x3_pp_kernel(const AttnParams p, int qtiles, int total_blocks, int nsplit, int serial) {
    static_assert(DH == 64 || DH == 32, "head widths of the reference: 256 / 4 and 128 / 4 channels");
    constexpr int NT = 512;
    constexpr int KROW = DH + 4;                 // K row: 32 floats of hi halves, 32 of lo halves, 4 pad
    constexpr int VROW = DH + 16;                // V row: same split, padded to 320 B (conflict-free transpose reads)
    constexpr int DT = DH / 32, KS = DH / 16;
    constexpr float SL2E = (DH == 64 ? 0.125f : 0.17677669529663687f) * LOG2E;       // 1/sqrt(DH) * log2(e)
    constexpr int LK = KT * DH / 4 / NT;         // float4 of K (and of V) per thread and tile: 2 / 1
    constexpr float P_SUM_LIMIT = 16384.f;       // per-lane partial row sum that forces a reference update
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool clk_on = p.clk_probe != nullptr && blockIdx.x == 0;      // workgroup-uniform
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    float* Ks = smem;                           // [4][KT][KROW]
    float* Vs = Ks + 4 * KT * KROW;             // [4][KT][VROW]   V stays key-major: the PV operand is read transposed
    float* Bs = Vs + 4 * KT * VROW;             // [4][KT]

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;                // 0: waves 0-3, 1: waves 4-7 (one phase behind)
    int id = xcd_remap(blockIdx.x, total_blocks);
    // key split: a unit's keys are cut into ns SHARES (a function of the unit's own counts), every share gives a partial result, the partials are merged in a fixed
    // order.  WHO computes the shares is the launch's choice and changes no bit: one workgroup per share (the grid carries nsplit = the launch's largest split; the last
    // workgroup of a unit to finish merges) when the launch would otherwise leave most of the chip idle, or - `serial` - ONE workgroup all shares of its unit, one after
    // the other, when the launch fills the chip without them (a ragged batch of four: parallel shares doubled its workgroups, 95 against 66 us per launch on the harder set)
    const int gs = serial ? 1 : nsplit;
    const int sp = id % gs; id /= gs;
    const int qt = id % qtiles; id /= qtiles;
    const int h = id % IMP_NUM_HEADS; id /= IMP_NUM_HEADS;
    const int sidx = id % p.nside;
    const int b = id / p.nside;
    const AttnSide& S = p.side[sidx];
    const int nq = imp_count(p.rc, S.qimg, b, S.nq), nk = imp_count(p.rc, S.kimg, b, S.nk);      // ragged batches: this pair's own counts; S.nq / S.nk = the padded layout
    const int q0 = qt * 256;
    if (q0 >= nq || nk <= 0) return;
    // the split of THIS (pair, side): a function of its own counts (imp_kernels.h attn_side_splits), whatever else the launch holds
    const int ns = nsplit > 1 ? attn_side_splits(nq, nk) : 1;
    if (sp >= ns) return;

    const int nt_all = (nk + KT - 1) / KT, t_per = (nt_all + ns - 1) / ns;
    int t0 = sp * t_per;                             // this share: tiles [t0, t0 + nt) of the keys
    int nt = min(nt_all, t0 + t_per) - t0;           // >= 1: the rule never makes more shares than it has tiles for
    const float* Qg = S.q + b * S.sq_b + h * DH;
    const float* Kg = S.k + b * S.sk_b + h * DH;
    const float* Vg = S.v + b * S.sk_b + h * DH;
    const uint8_t* mk = (MASKED && S.kmask) ? S.kmask + (long)b * S.nk : nullptr;      // (a launch without key masks is its own instantiation: no conditional mask load between the staged loads and their waits)

    // (round 4: the Q rows are only REQUESTED here; the first two K / V tiles are requested right behind them and the Q split waits
    // for its own loads alone - the prologue used to pay two dependent memory round trips with the matrix pipe idle, Q then K / V, while
    // all 256 workgroups of the launch pull their 64-KB Q tile and first tiles at once)
    f16x8 qh[KS], ql[KS];
    f32x4 qraw[KS][2];
    {
        const int qrow = q0 + wave * 32 + l31;
        const float* src = Qg + (long)(qrow < nq ? qrow : nq - 1) * p.ldq + 8 * half;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qraw[s][0] = *reinterpret_cast<const f32x4*>(src + 16 * s);
            qraw[s][1] = *reinterpret_cast<const f32x4*>(src + 16 * s + 4);
        }
    }

    // ---- staging: thread = 2 float4 of K and 2 of V (row f/16, channels 4(f%16)..) ---------------------------------
    const unsigned kv_bytes = (unsigned)(((long)(nk - 1) * p.ldk + DH) * 4);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, kv_bytes, 0x00020000);
    const int row_bytes = p.ldk * 4;
    int koff[LK];
#pragma unroll
    for (int j = 0; j < LK; ++j) {
        const int f = tid + j * NT;
        koff[j] = (f / (DH / 4)) * row_bytes + (f % (DH / 4)) * 16;
    }
    f32x4 rkA[LK], rkB[LK];                        // two staged tiles in flight (even / odd tile index)
    f32x4 rvA[LK], rvB[LK];
    unsigned char rbA = 1, rbB = 1;
    auto load_tile = [&](int t, f32x4 (&rk)[LK], f32x4 (&rv)[LK], unsigned char& rb) __attribute__((always_inline)) {
        const int k0 = (t0 + t) * KT;
        const int soff = k0 * row_bytes;
#pragma unroll
        for (int j = 0; j < LK; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsK, koff[j], soff, 0);
            rk[j] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
#pragma unroll
        for (int j = 0; j < LK; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsV, koff[j], soff, 0);
            rv[j] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
        if (tid < KT) {                       // key-validity byte, turned into the 0 / -inf bias when the tile is stored
            const int key = k0 + tid;
            unsigned char keep = key < nk;
            if (keep && mk) keep = mk[key];
            rb = keep;
        }
    };
    auto store_tile = [&](int slot, const f32x4 (&rk)[LK], const f32x4 (&rv)[LK], const unsigned char rb) __attribute__((always_inline)) {
        float* ks = Ks + slot * KT * KROW;
        float* vs = Vs + slot * KT * VROW;
#pragma unroll
        for (int j = 0; j < LK; ++j) {
            const int f = tid + j * NT;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            if (p.kv_planes) {                 // the rows already ARE the [hi | lo] image (written by the producer GEMM; 16-byte chunk c4 / 4 of it): plain copy
                *reinterpret_cast<f32x4*>(ks + row * KROW + c4) = rk[j];
                *reinterpret_cast<f32x4*>(vs + row * VROW + c4) = rv[j];
                continue;
            }
            u32x2 hi, lo;
            split4(rk[j], hi, lo);
            *reinterpret_cast<u32x2*>(ks + row * KROW + (c4 >> 1)) = hi;
            *reinterpret_cast<u32x2*>(ks + row * KROW + DH / 2 + (c4 >> 1)) = lo;
            split4(rv[j], hi, lo);
            *reinterpret_cast<u32x2*>(vs + row * VROW + (c4 >> 1)) = hi;
            *reinterpret_cast<u32x2*>(vs + row * VROW + DH / 2 + (c4 >> 1)) = lo;
        }
        if (tid < KT) Bs[slot * KT + tid] = rb ? 0.f : -INFINITY;
    };

    f32x16 oacc[DT], sacc[2];
    f16x8 ph[2][2], pl[2][2];
    float m_ref = 0.f;            // log2-domain reference exponent of this lane's query (identical in both lane halves)
    float l_run = 0.f;            // this lane's partial row sum, relative to m_ref
    bool need_slow = true;        // wave-uniform: some query of the wave has not seen an unmasked key yet

    // A operand of O^T += V^T . P^T through the LDS transpose read (ds_read_b64_tr_b16): every 16-lane group hands in
    // the addresses of a [4 keys][16 channels] block (lane i: key i/4, channels 4(i%4)..+3) and lane i receives channel
    // i of the 4 keys.  K-slot e of lane-half h is key 32 jb + 16 s2 + 4 h + 8 (e >> 2) + (e & 3) - the key order the
    // S^T accumulator registers (= the B operand P^T) already have - so two reads (keys +0..3 and +8..11) per fragment.
    const int vlane = ((4 * half + ((lane & 15) >> 2)) * VROW) * 4 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    // Fragment reads are software-pipelined by hand (the compiler otherwise issues the reads of a k-step only after
    // the previous step's MFMAs and waits for them in front of the next MFMA: ~150 idle pipe cycles per step): the
    // operands of step i+1 are read into the other half of a register double buffer before the MFMAs of step i are
    // issued; sched_barrier(0) pins that order.
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto read_v = [&](int slot, int g, f16x8 (&f)[4]) __attribute__((always_inline)) {            // f = {vh[0..DT), vl[0..DT)} of k-step g
        const char* vs = reinterpret_cast<const char*>(Vs + slot * KT * VROW) + vlane + (16 * g) * (VROW * 4);
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i) {
            const char* a = vs + (i % DT) * 64 + (i / DT) * (DH * 2);
            const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
            const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 8 * VROW * 4));
            f[i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto read_k = [&](int slot, int s, f16x8 (&f)[4]) __attribute__((always_inline)) {            // f = {kh[0], kh[1], kl[0], kl[1]} of k-step s
        const float* ks = Ks + slot * KT * KROW + l31 * KROW + 4 * half + 8 * s;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            f[i] = *reinterpret_cast<const f16x8*>(ks + (i & 1) * 32 * KROW + (i >> 1) * (DH / 2));
    };
    auto mfma_v = [&](int g, const f16x8 (&f)[4]) __attribute__((always_inline)) {
        const int jb = g >> 1, s2 = g & 1;
#pragma unroll
        for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[DT + d], ph[jb][s2], oacc[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DT; ++d)
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[d], pl[jb][s2], oacc[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[d], ph[jb][s2], oacc[d], 0, 0, 0);
    };
    // cneg = -m_ref in all 16 registers: the C operand of the FIRST MFMA of both S chains of a tile (the MFMA reads C from
    // cneg and writes D to sacc), so the accumulators need no 32 v_mov per tile; rewritten only when m_ref moves (slow path)
    f32x16 cneg;
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[r] = 0.f;
    auto mfma_k = [&](int s, const f16x8 (&f)[4]) __attribute__((always_inline)) {
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], qh[s], (s == 0) ? cneg : sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], qh[s], (s == 0) ? cneg : sacc[1], 0, 0, 0);
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], ql[s], sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], ql[s], sacc[1], 0, 0, 0);
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], qh[s], sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], qh[s], sacc[1], 0, 0, 0);
    };
    f16x8 fr[2][4];                                                 // fragment double buffer
    // PP_SPREAD (DH = 64): the operands of k-step i+1 are fetched one fragment at a time BETWEEN the MFMAs of step i (read order 2, 3, 0, 1 =
    // the order in which the MFMAs of a step first touch them) instead of as a block in front of them
    constexpr bool SPREAD = DT == 2;
    auto read_v1 = [&](int slot, int g, int i, f16x8& f) __attribute__((always_inline)) {
        const char* a = reinterpret_cast<const char*>(Vs + slot * KT * VROW) + vlane + (16 * g) * (VROW * 4) + (i % DT) * 64 + (i / DT) * (DH * 2);
        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
        const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 8 * VROW * 4));
        f = __builtin_bit_cast(f16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto read_k1 = [&](int slot, int s, int i, f16x8& f) __attribute__((always_inline)) {
        const float* ks = Ks + slot * KT * KROW + l31 * KROW + 4 * half + 8 * s;
        f = *reinterpret_cast<const f16x8*>(ks + (i & 1) * 32 * KROW + (i >> 1) * (DH / 2));
    };
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
    // O^T += V(slot)^T . P^T ; when kslot >= 0 the first K fragments of the following S^T are prefetched at the end; fr[0] already holds
    // the operands of step 0 when `have0`
    auto pv_mfmas = [&](int slot, int kslot, bool have0 = false) __attribute__((always_inline)) {
        if (!have0) read_v(slot, 0, fr[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (SPREAD) {
                const int jb = g >> 1, s2 = g & 1;
                f16x8 (&f)[4] = fr[g & 1];
                f16x8 (&n)[4] = fr[(g + 1) & 1];
                auto rd = [&](int i) { if (g < 3) read_v1(slot, g + 1, i, n[i]); else if (kslot >= 0) read_k1(kslot, 0, i, n[i]); };
                PP_SB();
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], ph[jb][s2], oacc[0], 0, 0, 0); PP_SB();
                rd(2); PP_SB();
                oacc[DT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], ph[jb][s2], oacc[DT - 1], 0, 0, 0); PP_SB();
                rd(3); PP_SB();
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], pl[jb][s2], oacc[0], 0, 0, 0); PP_SB();
                rd(0); PP_SB();
                oacc[DT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], pl[jb][s2], oacc[DT - 1], 0, 0, 0); PP_SB();
                rd(1); PP_SB();
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], ph[jb][s2], oacc[0], 0, 0, 0); PP_SB();
                oacc[DT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], ph[jb][s2], oacc[DT - 1], 0, 0, 0); PP_SB();
            } else {
                if (g < 3) read_v(slot, g + 1, fr[(g + 1) & 1]);
                else if (kslot >= 0) read_k(kslot, 0, fr[0]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_v(g, fr[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto qk_mfmas = [&](int kslot, bool prefetched) __attribute__((always_inline)) {
        if (!prefetched) read_k(kslot, 0, fr[0]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (SPREAD) {
                f16x8 (&f)[4] = fr[s & 1];
                f16x8 (&n)[4] = fr[(s + 1) & 1];
                auto rd = [&](int i) { if (s + 1 < KS) read_k1(kslot, s + 1, i, n[i]); };
                PP_SB();
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], qh[s], (s == 0) ? cneg : sacc[0], 0, 0, 0); PP_SB();
                rd(2); PP_SB();
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], qh[s], (s == 0) ? cneg : sacc[1], 0, 0, 0); PP_SB();
                rd(3); PP_SB();
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], ql[s], sacc[0], 0, 0, 0); PP_SB();
                rd(0); PP_SB();
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], ql[s], sacc[1], 0, 0, 0); PP_SB();
                rd(1); PP_SB();
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], qh[s], sacc[0], 0, 0, 0); PP_SB();
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], qh[s], sacc[1], 0, 0, 0); PP_SB();
            } else {
                if (s + 1 < KS) read_k(kslot, s + 1, fr[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_k(s, fr[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // P = exp2(acc - ref) for the 32 logits of this lane, split into the B-operand fragments; returns their sum
    auto probabilities = [&](float delta) __attribute__((always_inline)) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 ls2 = {0.f, 0.f};                 // two running sums: v_pk_add_f32 (one instruction per pair of probabilities)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = fast_exp2(sacc[jb][8 * s2 + e] - delta);
#pragma unroll
                for (int e = 0; e < 8; ++e) ls2[0] += pv[e];
                split8(pv, ph[jb][s2], pl[jb][s2]);
            }
        return ls2[0] + ls2[1];
    };

    auto tile_step = [&](int t, f32x4 (&rk)[LK], f32x4 (&rv)[LK], unsigned char& rb, f32x4 (&rk2)[LK], f32x4 (&rv2)[LK], unsigned char& rb2) __attribute__((always_inline)) {
        // =============================== X(t): matrix phase ===============================================
        __builtin_amdgcn_s_setprio(1);
        if (t > 0) pv_mfmas((t - 1) & 3, t & 3, true);
        qk_mfmas(t & 3, t > 0);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
        // =============================== Y(t): vector phase ===============================================
        if (mk != nullptr || (t0 + t + 1) * KT > nk) {
            const float* bs = Bs + (t & 3) * KT;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + jb * 32 + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sacc[jb][4 * g + e] += bias[e];
                }
        }
        float lsum = 0.f;
        bool slow = need_slow;
        const float base = 0.f;                 // the accumulators already hold S - m_ref
        if (!slow) {
            lsum = probabilities(base);
            bool big = !(lsum < P_SUM_LIMIT);                     // also catches inf / nan
            slow = __any(big);
        }
        if (slow) {
            float tmax = -INFINITY;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[jb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32)) - base;
            const float lq = l_run + __shfl_xor(l_run, 32);       // > 0 once the query has an anchored reference
            // anchored queries only ever raise their reference; an empty one takes the tile maximum as it is
            const float delta = (tmax == -INFINITY) ? 0.f : (lq > 0.f ? fmaxf(tmax, 0.f) : tmax);
            const float alpha = lq > 0.f ? fast_exp2(-delta) : 0.f;
            m_ref += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) cneg[r] = -m_ref;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            lsum = probabilities(base + delta);
            need_slow = __any(tmax == -INFINITY && !(lq > 0.f));
        }
        l_run += lsum;
        read_v(t & 3, 0, fr[0]);                                  // operands of the first k-step of X(t+1) (or of the final P.V): V(t) has been in LDS since phase 2t-2
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) store_tile((t + 2) & 3, rk, rv, rb);      // this set holds tile t+2 (same parity as t)
        if (t + 4 < nt) load_tile(t + 4, rk, rv, rb);
        PP_BARRIER();
    };
    constexpr int LDO = DH + 1;
    // scratch of a split unit: floats per partial: O [256][DH] | m [256] | l [256]
    constexpr int PART = 256 * DH + 512;
    // rows staged with a 16-byte aligned pitch: 128-bit write-through stores, LPR lanes per row, RPI rows per instruction
    constexpr int LDP = DH + 4, LPR = DH / 4, RPI = 64 / LPR;
    float l_tot = 0.f;
    const int sp_end = serial ? ns : sp + 1;
    // (the first share's first tiles are requested in front of the Q split - round 4: the Q rows were only REQUESTED above - so that the split waits for
    // its own loads alone; the raw Q registers are dead before the loop over the shares begins)
    load_tile(0, rkA, rvA, rbA);
    if (nt > 1) load_tile(1, rkB, rvB, rbB);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const f32x4 a = qraw[s][0], c = qraw[s][1];
        const float x[8] = {a[0] * SL2E, a[1] * SL2E, a[2] * SL2E, a[3] * SL2E,
                            c[0] * SL2E, c[1] * SL2E, c[2] * SL2E, c[3] * SL2E};
        split8(x, qh[s], ql[s]);
    }
    for (int spi = sp; spi < sp_end; ++spi) {
        if (spi != sp) {
            t0 = spi * t_per;
            nt = min(nt_all, t0 + t_per) - t0;
            load_tile(0, rkA, rvA, rbA);
            if (nt > 1) load_tile(1, rkB, rvB, rbB);
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cneg[r] = 0.f;
        m_ref = 0.f; l_run = 0.f; need_slow = true;
        store_tile(0, rkA, rvA, rbA);
        if (nt > 2) load_tile(2, rkA, rvA, rbA);
        if (nt > 1) store_tile(1, rkB, rvB, rbB);
        if (nt > 3) load_tile(3, rkB, rvB, rbB);
        PP_BARRIER();
        if (group == 1) PP_BARRIER();
        for (int t = 0; t < nt; t += 2) {
            tile_step(t, rkA, rvA, rbA, rkB, rvB, rbB);
            if (t + 1 < nt) tile_step(t + 1, rkB, rvB, rbB, rkA, rvA, rbA);
        }
        pv_mfmas((nt - 1) & 3, -1, true);
        if (group == 0) PP_BARRIER();               // balance the extra barrier of waves 4-7
        PP_BARRIER();                               // everyone is done with the ring: reuse it for the transposition
        l_tot = l_run + __shfl_xor(l_run, 32);
        if (ns > 1) {
            // ---- this share's PARTIAL result (O relative to its own m_ref, l, m_ref) goes to scratch with agent-coherent write-through stores
            const long unit = (((long)b * p.nside + sidx) * IMP_NUM_HEADS + h) * qtiles + qt;
            const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.split_ws + unit * nsplit * PART), 0, (unsigned)(nsplit * PART * 4), 0x00020000);
            float* otp = smem + wave * 32 * LDP;
            const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) otp[l31 * LDP + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = oacc[d][r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 32 / RPI; ++j) {
                const int qi = j * RPI + prow;
                const f32x4 v = *reinterpret_cast<const f32x4*>(otp + qi * LDP + pc4);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                       rsW, ((spi * PART) + (wave * 32 + qi) * DH + pc4) * 4, 0, 16);
            }
            if (half == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m_ref), rsW, (spi * PART + 256 * DH + wave * 32 + l31) * 4, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l_tot), rsW, (spi * PART + 256 * DH + 256 + wave * 32 + l31) * 4, 0, 16);
            }
            __builtin_amdgcn_s_waitcnt(0);                         // the write-through stores are acknowledged
            __syncthreads();                                       // ... and nobody reads the staging rows any more (a next share refills the ring)
        }
    }
    if (ns > 1) {
        // ---- merge: m = max m_s, O = sum_s 2^(m_s - m) O_s, l likewise, out = O / l - by the workgroup that computed all shares (serial), or by the LAST of the
        // unit's workgroups to arrive: a ticket per (pair, side, head, query tile), nobody waits for anybody (no spin): the ticket is a relaxed agent-scope atomic
        // taken after the stores are acknowledged
        const long unit = (((long)b * p.nside + sidx) * IMP_NUM_HEADS + h) * qtiles + qt;
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.split_ws + unit * nsplit * PART), 0, (unsigned)(nsplit * PART * 4), 0x00020000);
        float* ot = smem + wave * 32 * LDO;
        const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
        __shared__ int s_last;
        if (tid == 0) {
            if (serial) s_last = 1;
            else {
                const unsigned old = __hip_atomic_fetch_add(p.split_cnt + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = old == (unsigned)ns - 1;
                if (s_last) __hip_atomic_store(p.split_cnt + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
            }
        }
        __syncthreads();
        if (!s_last) return;
        // merge (this wave: its 32 queries; lane l31 owns query l31 for the scalars)
        float* wq = smem + 8 * 32 * LDP + wave * 32 * 8;       // [32 queries][up to 7 split weights | total l], behind the staging rows
        {
            const int moff = (256 * DH + wave * 32 + l31) * 4, loff = moff + 256 * 4;
            float mmax = -INFINITY;
            for (int s2 = 0; s2 < ns; ++s2) {
                const float ms = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, s2 * PART * 4 + moff, 0, 16));
                const float ls = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, s2 * PART * 4 + loff, 0, 16));
                if (ls > 0.f) mmax = fmaxf(mmax, ms);
            }
            float L = 0.f;
            for (int s2 = 0; s2 < ns; ++s2) {
                const float ms = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, s2 * PART * 4 + moff, 0, 16));
                const float ls = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsW, s2 * PART * 4 + loff, 0, 16));
                const float w = ls > 0.f ? fast_exp2(ms - mmax) : 0.f;
                L = fmaf(w, ls, L);
                if (half == 0) wq[l31 * 8 + s2] = w;
            }
            if (half == 0) {
                wq[l31 * 8 + 7] = L;                          // (nsplit <= 7)
                const int qrow = q0 + wave * 32 + l31;
                if (S.lse && qrow < nq) S.lse[((long)b * IMP_NUM_HEADS + h) * S.nq + qrow] = mmax * (1.0f / LOG2E) + logf(L);
            }
        }
        __syncthreads();
        // rows: all partial loads of a 16-row pass are issued before any of them is used
        for (int pass = 0; pass < 2; ++pass) {
            constexpr int NJ = 16 / RPI;
            f32x4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s2 = 0; s2 < ns; ++s2) {
                u32x4 v[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ((s2 * PART) + (wave * 32 + pass * 16 + j * RPI + prow) * DH + pc4) * 4, 0, 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float w = wq[(pass * 16 + j * RPI + prow) * 8 + s2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(w, __uint_as_float(v[j][e]), acc[j][e]);
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int qi = pass * 16 + j * RPI + prow;
                const float inv = 1.0f / wq[qi * 8 + 7];
#pragma unroll
                for (int e = 0; e < 4; ++e) ot[qi * LDO + pc4 + e] = acc[j][e] * inv;
            }
        }
        __syncthreads();
        store_attention_rows<DH>(p, S, b, h, q0 + wave * 32, nq, ot, LDO, lane);
        return;
    }
    {
        // round 4: the wave's 32 output rows leave as 16-byte stores of full 256-byte rows (4 rows per instruction; 4-byte stores, one row
        // per instruction, before): all 256 workgroups write their 64-KB tile at the same moment with the matrix pipe idle
        float* otp = smem + wave * 32 * LDP;
        const float inv_l = 1.0f / l_tot;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                // O / l as one division per lane and Markstein's correction per element: bit-identical to a / l (imp_kernels.h imp_div_by; tools/probe/div3_probe.hip)
                otp[l31 * LDP + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = imp_div_by(oacc[d][r], l_tot, inv_l);
            }
        if (S.lse && half == 0) {
            const int qrow = q0 + wave * 32 + l31;
            if (qrow < nq) S.lse[((long)b * IMP_NUM_HEADS + h) * S.nq + qrow] = m_ref * (1.0f / LOG2E) + logf(l_tot);
        }
        __syncthreads();
        const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
        float* Og = S.out + b * S.so_b + h * DH;
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j) {
            const int qi = j * RPI + prow;
            const int qrow = q0 + wave * 32 + qi;
            const f32x4 v = *reinterpret_cast<const f32x4*>(otp + qi * LDP + pc4);
            if (qrow < nq) *reinterpret_cast<f32x4*>(Og + (long)qrow * p.ldo + pc4) = v;
        }
    }
    if (clk_on && threadIdx.x == 0) {           // (wave 0 of workgroup 0: the others finish within a few hundred cycles of it)
        __builtin_amdgcn_s_waitcnt(0);
        p.clk_probe[0] += __builtin_readcyclecounter() - clk_c0;
        p.clk_probe[1] += __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}

template <int DH>
hipError_t launch_pp(const AttnParams& p, int batch, int maxq, int nsplit, int serial, hipStream_t stream) {
    const int qtiles = (maxq + 255) / 256;
    const int total = qtiles * IMP_NUM_HEADS * p.nside * batch * (serial ? 1 : nsplit);
    const size_t lds = (size_t)(4 * KT * (DH + 4) + 4 * KT * (DH + 16) + 4 * KT) * sizeof(float);
    const bool masked = p.side[0].kmask != nullptr || (p.nside == 2 && p.side[1].kmask != nullptr);
    if (masked) {
        if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_pp_kernel<DH, true>, lds)) return e;
        hipLaunchKernelGGL((attn_f16x3_pp_kernel<DH, true>), dim3(total), dim3(512), lds, stream, p, qtiles, total, nsplit, serial);
    } else {
        if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_pp_kernel<DH, false>, lds)) return e;
        hipLaunchKernelGGL((attn_f16x3_pp_kernel<DH, false>), dim3(total), dim3(512), lds, stream, p, qtiles, total, nsplit, serial);
    }
    return hipGetLastError();
}

}  // namespace

// Key split of the ping-pong kernel for (pair, side) units that would leave most of the chip idle on their own (one pair of ~1000 keypoints =
// 32 workgroups on 256 CUs, each walking serially over all keys: 37 us at N = 1024 however the queries are tiled, tools/probe/attn_small.py):
// ns workgroups share a query tile's keys and the last one to finish merges the partials.  Round 6: the split of a unit is a function of the
// unit's OWN query / key counts (attn_side_splits, imp_kernels.h) - a split changes the order in which a query's keys are summed, so a rule that
// looked at the launch (batch size, the other pairs) made a pair's result depend on the batch it travelled in.  The launch
// is sized for the largest split of its units; workgroups beyond a unit's own split leave at once.
// Needs the scratch of AttnParams (split_ws / split_cnt); without it nothing is split.
int attention_f16x3_splits(const AttnParams& p, int batch) {
    if (!p.split_ws || !p.split_cnt) return 1;
    int s = 1;
    for (int sd = 0; sd < p.nside; ++sd)
        for (int b = 0; b < (p.rc.on ? batch : 1); ++b) {
            const int nq = p.rc.on ? p.rc.n[p.side[sd].qimg][b] : p.side[sd].nq, nk = p.rc.on ? p.rc.n[p.side[sd].kimg][b] : p.side[sd].nk;
            if (nq <= 0 || nk <= 0) continue;                       // retired pair
            const int u = attn_side_splits(nq, nk);
            if (u > s) s = u;
        }
    return s;
}
size_t attention_f16x3_split_floats(const AttnParams& p, int batch, int nsplit) {
    int maxq = p.side[0].nq;
    if (p.nside == 2 && p.side[1].nq > maxq) maxq = p.side[1].nq;
    return (size_t)((maxq + 255) / 256) * IMP_NUM_HEADS * p.nside * batch * nsplit * (256 * (size_t)p.dh + 512);
}
size_t attention_f16x3_split_units(const AttnParams& p, int batch) {
    int maxq = p.side[0].nq;
    if (p.nside == 2 && p.side[1].nq > maxq) maxq = p.side[1].nq;
    return (size_t)((maxq + 255) / 256) * IMP_NUM_HEADS * p.nside * batch;
}

// the inverse for readers that want fp32 (pooling's column sums, probability materialisation): x' = hi + lo, exact in fp32 (22 bits)
__global__ __launch_bounds__(256) void attn_kv_unplanes_kernel(const float* base, long rows, int ld, int col0, int dh, float* out, int ldo) {
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (unit >= rows * IMP_NUM_HEADS) return;
    const long row = unit / IMP_NUM_HEADS;
    const int h = (int)(unit % IMP_NUM_HEADS);
    const _Float16* seg = reinterpret_cast<const _Float16*>(base + row * ld + col0 + h * dh);
    if (lane < dh) out[row * ldo + h * dh + lane] = (float)seg[lane] + (float)seg[dh + lane];
}

hipError_t launch_attn_kv_unplanes(const float* base, long rows, int ld, int col0, int dh, float* out, int ldo, hipStream_t stream) {
    const long units = rows * IMP_NUM_HEADS;
    hipLaunchKernelGGL(attn_kv_unplanes_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, base, rows, ld, col0, dh, out, ldo);
    return hipGetLastError();
}

hipError_t launch_attention_f16x3(const AttnParams& p, int batch, hipStream_t stream) {
    int maxq = p.side[0].nq;
    if (p.nside == 2 && p.side[1].nq > maxq) maxq = p.side[1].nq;
    if (maxq <= 0 || batch <= 0) return hipSuccess;
    if (p.dh != 64 && p.dh != 32) return hipErrorInvalidValue;
    // Round 6: ONE kernel for every size - the phase-staggered 256-query kernel (rows past nq / nk are clamped / masked).  Rounds 1-5 sent launches
    // whose largest side had <= 192 queries to the lock-step kernels (attn_f16x3_kernel, removed): a choice per LAUNCH, so a small pair took another
    // kernel - another order of summation - beside a large pair than alone.  What a small pair loses (a few microseconds at D = 128 or
    // option kv_image = 0; split-half K / V images always ran here) is the price of results that depend on the pair alone.
    const int nsplit = attention_f16x3_splits(p, batch);
    // who computes the shares of a split unit (no bit depends on it): one workgroup each while all of them fit the chip at once or the units alone would leave
    // more than half of it idle, else one workgroup per query tile all of its shares in turn
    int serial = 0;
    if (nsplit > 1) {
        int live = 0, shares = 0;                  // workgroups with work: one per query tile and head / one per share of it
        for (int sd = 0; sd < p.nside; ++sd)
            for (int b = 0; b < batch; ++b) {
                const int nq = p.rc.on ? p.rc.n[p.side[sd].qimg][b] : p.side[sd].nq, nk = p.rc.on ? p.rc.n[p.side[sd].kimg][b] : p.side[sd].nk;
                if (nq <= 0 || nk <= 0) continue;
                live += ((nq + 255) / 256) * IMP_NUM_HEADS;
                shares += ((nq + 255) / 256) * IMP_NUM_HEADS * attn_side_splits(nq, nk);
            }
        serial = shares > 256 && live >= 128;      // the shares do not fit the chip at once and the units alone fill half of it
        if (p.share_mode) serial = p.share_mode == 2;
    }
    return p.dh == 64 ? launch_pp<64>(p, batch, maxq, nsplit, serial, stream) : launch_pp<32>(p, batch, maxq, nsplit, serial, stream);
}
