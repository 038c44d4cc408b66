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
#ifdef PP_PROFILE     // tools/probe/attn_probe.hip: per-wave cycle counts of the phases of workgroup 0
__device__ unsigned long long pp_prof[8][8];
__device__ unsigned long long pp_span[8][3];    // per wave of workgroup 0: prologue, key loop, epilogue
#define PP_CLK(i) { const unsigned long long c_ = __builtin_readcyclecounter(); prof[i] += c_ - tlast; tlast = c_; }
#else
#define PP_CLK(i)
#endif
#ifdef PP_TIMELINE    // absolute s_memtime stamps of every wave of workgroup 0 over key tiles [PP_TL_T0, PP_TL_T0 + 4): the per-SIMD timeline
#define PP_TL_T0 8
__device__ unsigned long long pp_tl[8][4][8];
__device__ unsigned pp_hwid[8];
#define PP_TL(i) { if (tl_on) { const unsigned long long c_ = __builtin_readcyclecounter(); if (lane == 0) pp_tl[wave][t - PP_TL_T0][i] = c_; } }
#else
#define PP_TL(i)
#endif
#ifndef PP_INIT_IN_ACC
#define PP_INIT_IN_ACC 1    // S accumulators start at -m_ref (1) or at 0 with the reference subtracted in the vector phase (0)
#endif
#ifndef PP_CNEG
#define PP_CNEG 1           // first MFMA of each S chain reads its C operand from a constant -m_ref vector instead of 32 v_mov per tile (A/B on one box: 94.3 vs 95.4 us)
#endif
#ifndef PP_PKADD
#define PP_PKADD 0          // row sums with v_pk_add_f32: measured 118 vs 95 us (the register pairing costs more than the adds save)
#endif
#ifndef PP_P1
#define PP_P1 0             // EXPERIMENT (not the product): probabilities enter P.V as ONE half (2 MFMAs per product instead of 3 there);
#endif                      // DESIGN.md section 4 has what it buys and what it costs in score accuracy
#ifndef PP_SPREAD
#define PP_SPREAD 1         // operand reads of k-step i+1 issued BETWEEN the MFMAs of step i (1) instead of as a block in front of them (0)
#endif
#ifndef PP_XSM
#define PP_XSM 0            // probability quarters (16 keys x 32 queries each, of the 4 per tile) whose exp2 / row sum / hi-lo split is NOT done in the vector
#endif                      // phase but between the MFMAs of the NEXT matrix phase's P.V (needs PP_SPREAD): the matrix wave has idle issue slots, the vector wave is issue-bound
#ifndef PP_STRAIGHT
#define PP_STRAIGHT 0       // 1: the staged tiles are requested unconditionally (tiles past the end read as zeros through the descriptor's bounds check), so
#endif                      // the compiler KNOWS a younger tile's four loads are in flight and waits with vmcnt(7..4) instead of the conservative vmcnt(3..0)
#ifndef PP_WHATIF
#define PP_WHATIF 0         // TIMING EXPERIMENTS ONLY (wrong results): 1 = no staging inside the key loop (the ring keeps its first tiles), 2 = no hi / lo
#endif                      // split of P (pl = ph), 3 = both: what removing that work from the vector phase could buy at most
#ifndef PP_DIV3
#define PP_DIV3 1           // epilogue normalisation O / l as ONE IEEE division y = 1 / l per lane and, per element, q0 = a y; r = fma(-q0, l, a); q = fma(r, y, q0)
#endif                      // (Markstein's correction: BIT-IDENTICAL to a / l - tools/probe/div3_probe.hip: 0 mismatches in 5.5e11 random quotients, the bare product a y differs in 27 %)
#ifndef PP_RCP
#define PP_RCP 0            // epilogue normalisation: 32 IEEE divisions per lane (0) or one division and 32 multiplications (1: -1 % per launch, results move by <= 1 ulp - enough to flip a knife-edge mutual-nearest-neighbour decision of the fixture ragged_dgnns_l15_b4, so the product keeps the divisions)
#endif
#ifndef PP_PREFETCH
#define PP_PREFETCH 1       // the first V fragments of the next matrix phase are read at the end of the vector phase (before the barrier)
#endif
#ifndef PP_PRIO
#define PP_PRIO 0           // 0: priority 1 around every matrix phase; 1: no priorities; 2: waves 4-7 at priority 1 for the whole loop
#endif
#ifndef PP_DMA_DEFAULT
#define PP_DMA_DEFAULT 0    // the LDS-DMA staging variant of the kernel (template parameter DMA) as the default for split-half K / V images at DH = 64; IMP_ATTN_DMA overrides
#endif
#ifndef PP_DMA_SPREAD
#define PP_DMA_SPREAD 1     // where a wave requests its pieces: 1 = every wave requests tile t + 2 inside X(t), one piece behind the MFMAs of each k-step of K.Q^T (a piece costs ~20 issue
#endif                      // cycles + SALU: at the head of the phase it delays the MFMA stream of the pole wave); 0 = waves 0-3 at the head of X(t) (tile t + 2), waves 4-7 at the head of Y(t) (tile t + 3)
#ifndef PP_LOADS_IN_X
#define PP_LOADS_IN_X 0    // where the global loads of the staged tile are issued: matrix phase (1) or vector phase (0); measured equal
#endif

// DMA (round 5, DH = 64, split-half K / V images only): the tiles of the ring are filled by LDS-DMA (global_load_lds_dwordx4: global -> LDS without a
// register round trip) instead of 4 buffer loads + 4 ds_write_b128 per thread and tile.  An LDS-DMA instruction writes 64 x 16 bytes LANE-LINEAR at M0, so the
// row-pitched image of a tile (K 64 x 272 B = 17 KB, V 64 x 320 B = 20 KB: the pads stay, every fragment read is unchanged) is cut into 1-KB PIECES and a
// lane fetches whatever 16-byte chunk belongs at its position of the piece (a lane that lands in a pad fetches a neighbouring chunk: never read).  37 pieces
// per tile: every wave requests five consecutive ones (waves 0-3 of K, waves 4-7 of V; three K pieces twice).  Schedule, in barrier phases (phase p = between the
// p-th and the (p + 1)-th barrier; waves 0-3: X(t) = 2t, Y(t) = 2t + 1, waves 4-7 one later): a wave requests tile T inside its X(T - 2), one piece behind each
// k-step of K.Q^T - phase 2T - 4 for waves 0-3, 2T - 3 for waves 4-7 (PP_DMA_SPREAD = 0: waves 0-3 at the head of X(T - 2), waves 4-7 at the head of Y(T - 3), phase
// 2T - 4 for both); the last reads of the slot's previous occupant, V(T - 4), retired with the lgkmcnt(0) of the barrier that ends phase 2T - 5.  Every wave waits for
// its own pieces of T (counted vmcnt: the pieces of T + 1 stay in flight) before the barrier that ends phase 2T - 1; the first read of T is K(T) in phase 2T.  The
// compiler does not see the DMA (inline asm: hipcc would drain a DMA it knows about with vmcnt(0) before the next LDS read), so every wait is written here.
// Measured (profiles/r05/MEASURED.md): bit-identical, 21 VGPRs fewer, -2.6 % cycles per phase, equal in time (the launch is power-limited): off by default.
template <int DH, bool MASKED, bool DMA = false>
__global__ __launch_bounds__(512, 2) void attn_f16x3_pp_kernel(const AttnParams p, int qtiles, int total_blocks, int nsplit) {
    static_assert(DH == 64 || DH == 32, "head widths of the reference: 256 / 4 and 128 / 4 channels");
    static_assert(!DMA || (DH == 64 && !PP_STRAIGHT && !PP_LOADS_IN_X && !PP_WHATIF), "the LDS-DMA staging is written for the product's configuration");
    constexpr int NT = 512;
    constexpr int KROW = DH + 4;                 // K row: 32 floats of hi halves, 32 of lo halves, 4 pad
    constexpr int VROW = DH + 16;                // V row: same split, padded to 320 B (conflict-free transpose reads)
    constexpr int DT = DH / 32, KS = DH / 16;
    constexpr float SL2E = (DH == 64 ? 0.125f : 0.17677669529663687f) * LOG2E;       // 1/sqrt(DH) * log2(e)
    constexpr int LK = KT * DH / 4 / NT;         // float4 of K (and of V) per thread and tile: 2 / 1
    constexpr float P_SUM_LIMIT = 16384.f;       // per-lane partial row sum that forces a reference update
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef PP_PROFILE
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
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
    const int sp = id % nsplit; id /= nsplit;   // key split: this workgroup walks over tiles [t0, t0 + nt) of the keys (nsplit = the launch's largest split)
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
    const int t0 = sp * t_per;
    const int nt = min(nt_all, t0 + t_per) - t0;     // >= 1: the launcher never makes more splits than it has tiles for
    // (DMA) the same three as scalars for the request logic: a ragged batch's nk comes out of a memory load (imp_count), so everything derived from it is "divergent" to
    // the compiler - conditions become lane masks, addresses land in VGPRs - although it is uniform
    [[maybe_unused]] const int dnk = DMA ? __builtin_amdgcn_readfirstlane(nk) : nk, dnt = DMA ? __builtin_amdgcn_readfirstlane(nt) : nt, dt0 = DMA ? __builtin_amdgcn_readfirstlane(t0) : t0;
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
        if constexpr (!DMA) {
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
        }
#if PP_STRAIGHT
        {                                     // no per-lane branch (every wave computes it, wave 0 uses it); the mask byte is loaded only by masked launches
            const int key = k0 + (tid & (KT - 1));
            unsigned char keep = key < nk;
            if (mk != nullptr) { const unsigned char m = mk[key < nk ? key : 0]; keep = keep ? m : 0; }
            rb = keep;
        }
#else
        if (tid < KT) {                       // key-validity byte, turned into the 0 / -inf bias when the tile is stored
            const int key = k0 + tid;
            unsigned char keep = key < nk;
            if (keep && mk) keep = mk[key];
            rb = keep;
        }
#endif
    };
    auto store_tile = [&](int slot, const f32x4 (&rk)[LK], const f32x4 (&rv)[LK], const unsigned char rb) __attribute__((always_inline)) {
        if constexpr (!DMA) {
            float* ks = Ks + slot * KT * KROW;
            float* vs = Vs + slot * KT * VROW;
#pragma unroll
            for (int j = 0; j < LK; ++j) {
                const int f = tid + j * NT;
                const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
                if (p.kv_planes) {                 // EXPERIMENT: the rows already ARE the [hi | lo] image (16-byte chunk c4 / 4 of it): plain copy
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
        }
        if (tid < KT) Bs[slot * KT + tid] = rb ? 0.f : -INFINITY;
    };

    // ---- DMA staging: this wave's 5 CONSECUTIVE pieces of every tile: waves 0-3 K pieces 0-4, 5-9, 10-14, 12-16 (three requested twice), waves 4-7 V pieces 0-4 .. 15-19.
    // Consecutive, so that ONE M0 value and ONE scalar base serve the five requests of a tile: request j carries the immediate offset (j - 2) KB, which the hardware adds to
    // the LDS AND to the global address, the lane's source offset takes it back (+ 2 KB, so that it stays positive; the scalar base is 2 KB low).  A request then is ONE
    // instruction - the first version computed a base and an M0 per piece, ~9 scalar instructions each, and measured +400 cycles on the matrix phase it sat in
    constexpr int DMA_NP = 5;                                          // pieces per wave and tile
    constexpr int DMA_KP = KT * KROW * 4 / 1024, DMA_VP = KT * VROW * 4 / 1024;      // 17 + 20 pieces of 1 KB (DH = 64)
    static_assert(!DMA || ((KT * KROW * 4) % 1024 == 0 && (KT * VROW * 4) % 1024 == 0 && DMA_KP > 3 * DMA_NP && DMA_KP <= 4 * DMA_NP && DMA_VP == 4 * DMA_NP),
                  "a tile image is a whole number of 1-KB pieces; four waves x 5 consecutive pieces cover K, four cover V");
    [[maybe_unused]] unsigned dma_voff[DMA_NP];                        // per lane: row * row_bytes + 16 * chunk of its position in piece j, + 4096 - 1024 j
    [[maybe_unused]] const char* dma_base = nullptr;                   // wave-uniform: K or V of this (pair, head), 2 KB low
    [[maybe_unused]] unsigned dma_m0 = 0, dma_slot = 0;                // wave-uniform: LDS byte address of the wave's third piece in slot 0; bytes per slot
    [[maybe_unused]] const bool dma_isk = wave < 4;
    [[maybe_unused]] const int dma_p0 = dma_isk ? min(wave * DMA_NP, DMA_KP - DMA_NP) : (wave - 4) * DMA_NP;      // the wave's first piece
    // row and 16-byte chunk of this lane's position in piece j of the wave (a position inside a row's pad takes the row's last chunk: never read)
    auto dma_pos = [&](int j, int& r, int& c) __attribute__((always_inline)) {
        const int pitch = dma_isk ? KROW * 4 : VROW * 4;
        const int P = 1024 * (dma_p0 + j) + 16 * lane;
        r = P / pitch;
        c = min((P - r * pitch) >> 4, DH / 4 - 1);
    };
    [[maybe_unused]] const unsigned dma_tile_bytes = __builtin_amdgcn_readfirstlane((unsigned)(KT * row_bytes));      // bytes of global memory per key tile
    if constexpr (DMA) {
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned ks0 = (unsigned)(size_t)(lds_char*)reinterpret_cast<char*>(Ks), vs0 = (unsigned)(size_t)(lds_char*)reinterpret_cast<char*>(Vs);
#pragma unroll
        for (int j = 0; j < DMA_NP; ++j) {
            int r, c;
            dma_pos(j, r, c);
            dma_voff[j] = (unsigned)(r * row_bytes + 16 * c + 4096 - 1024 * j);
        }
        // (wave-uniform values pinned to scalar registers: a base that the compiler keeps in VGPRs cannot be the scalar address operand of the requests)
        const unsigned long long b64 = reinterpret_cast<unsigned long long>(dma_isk ? Kg : Vg) - 2048ull;
        dma_base = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64));
        dma_m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((dma_isk ? ks0 : vs0) + 1024u * (unsigned)dma_p0 + 2048u));
        dma_slot = dma_isk ? KT * KROW * 4 : KT * VROW * 4;
    }
    // the requests of tile t (relative to t0) into its ring slot, as three steps so that they can sit in different gaps of an MFMA stream: dma_open (scalar set-up, M0, piece 0),
    // dma_piece (1 .. 3), dma_close (piece 4).  M0 is written and left: hipcc sets M0 itself before any use it makes of it and makes none in this kernel (no m0 in the ISA of the
    // register-staged instantiations: `grep -c m0` on the -save-temps .s), so between open and close it is ours.
    // Rows past nk (the last tile of a ragged key count) repeat row nk - 1: finite bytes under a -inf bias
    [[maybe_unused]] const char* dma_src = nullptr;
#define PP_GLDS(OFF) "global_load_lds_dwordx4 %0, %1 offset:" #OFF
    // (the three steps assume a FULL tile: all 64 keys exist; the caller sends a partial last tile through dma_issue)
    auto dma_open = [&](int t) __attribute__((always_inline)) {
        if constexpr (DMA) {
            dma_src = dma_base + (unsigned)(dt0 + t) * dma_tile_bytes;  // (a 32-bit scalar product: a 64-bit or a VALU product would put the address into VGPRs)
            const unsigned m0v = dma_m0 + (unsigned)(t & 3) * dma_slot;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048" :: "v"(dma_voff[0]), "s"(dma_src), "s"(m0v) : "memory");
        }
    };
    auto dma_piece = [&](int j) __attribute__((always_inline)) {      // j = 1, 2, 3 (compile time after unrolling)
        if constexpr (DMA) {
            if (j == 1) asm volatile(PP_GLDS(-1024) :: "v"(dma_voff[1]), "s"(dma_src) : "memory");
            else if (j == 2) asm volatile(PP_GLDS(0) :: "v"(dma_voff[2]), "s"(dma_src) : "memory");
            else asm volatile(PP_GLDS(1024) :: "v"(dma_voff[3]), "s"(dma_src) : "memory");
        }
    };
    auto dma_close = [&]() __attribute__((always_inline)) {
        if constexpr (DMA) asm volatile(PP_GLDS(2048) :: "v"(dma_voff[4]), "s"(dma_src) : "memory");
    };
#undef PP_GLDS
    // all five requests of tile t in one place; this is also the path of a partial last tile (its missing rows repeat row nk - 1)
    auto dma_issue = [&](int t) __attribute__((always_inline)) {
        if constexpr (DMA) {
            const int k0 = (dt0 + t) * KT;
            if (k0 + KT > dnk) {                                       // (workgroup-uniform; at most one tile per workgroup)
                const char* src = dma_base + (unsigned)(dt0 + t) * dma_tile_bytes;
                const unsigned m0v = dma_m0 + (unsigned)(t & 3) * dma_slot;
                unsigned vo[DMA_NP];
#pragma unroll
                for (int j = 0; j < DMA_NP; ++j) {
                    int r, c;
                    dma_pos(j, r, c);
                    vo[j] = (unsigned)(min(r, dnk - 1 - k0) * row_bytes + 16 * c + 4096 - 1024 * j);
                }
                asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %5 offset:-2048\n\tglobal_load_lds_dwordx4 %1, %5 offset:-1024\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                             "global_load_lds_dwordx4 %3, %5 offset:1024\n\tglobal_load_lds_dwordx4 %4, %5 offset:2048"
                             :: "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(vo[4]), "s"(src), "s"(m0v) : "memory");
            } else {
                dma_open(t);
#pragma unroll
                for (int j = 1; j < DMA_NP - 1; ++j) dma_piece(j);
                dma_close();
            }
        }
    };
    // this wave's pieces of every tile it requested - but, `younger`, of the last one - have landed
    auto dma_wait = [&](bool younger) __attribute__((always_inline)) {
        if constexpr (DMA) {
            if (younger) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    f32x16 oacc[DT], sacc[2];
    f16x8 ph[2][2], pl[2][2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_ref = 0.f;            // log2-domain reference exponent of this lane's query (identical in both lane halves)
    float l_run = 0.f;            // this lane's partial row sum, relative to m_ref
    bool need_slow = true;        // wave-uniform: some query of the wave has not seen an unmasked key yet
    load_tile(0, rkA, rvA, rbA);
    if (PP_STRAIGHT || nt > 1) load_tile(1, rkB, rvB, rbB);
    dma_issue(0);                                   // (DMA: the two tiles travel while Q is split; the registers above then carry the key-validity bytes only)
    if (dnt > 1) dma_issue(1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const f32x4 a = qraw[s][0], c = qraw[s][1];
        const float x[8] = {a[0] * SL2E, a[1] * SL2E, a[2] * SL2E, a[3] * SL2E,
                            c[0] * SL2E, c[1] * SL2E, c[2] * SL2E, c[3] * SL2E};
        split8(x, qh[s], ql[s]);
    }
    store_tile(0, rkA, rvA, rbA);
    if (PP_STRAIGHT || nt > 2) load_tile(2, rkA, rvA, rbA);
    if (nt > 1) store_tile(1, rkB, rvB, rbB);
#if !PP_LOADS_IN_X
    if (PP_STRAIGHT || nt > 3) load_tile(3, rkB, rvB, rbB);
#endif
    dma_wait(false);                                // tiles 0 and 1 are in the ring
    PP_BARRIER();
    if (group == 1) PP_BARRIER();
    if (DMA && !PP_DMA_SPREAD && group == 1 && dnt > 2) dma_issue(2);      // barrier phase 0 (waves 0-3 issue it at the head of X(0))
#if PP_PRIO == 2
    if (group == 1) __builtin_amdgcn_s_setprio(1);
#endif

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
            if (!PP_P1) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[d], pl[jb][s2], oacc[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[d], ph[jb][s2], oacc[d], 0, 0, 0);
    };
    // cneg = -m_ref in all 16 registers: the C operand of the FIRST MFMA of both S chains of a tile (the MFMA reads C from
    // cneg and writes D to sacc), so the accumulators need no 32 v_mov per tile; rewritten only when m_ref moves (slow path)
    f32x16 cneg;
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[r] = 0.f;
    auto mfma_k = [&](int s, const f16x8 (&f)[4]) __attribute__((always_inline)) {
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], qh[s], (PP_CNEG && PP_INIT_IN_ACC && s == 0) ? cneg : sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], qh[s], (PP_CNEG && PP_INIT_IN_ACC && s == 0) ? cneg : sacc[1], 0, 0, 0);
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], ql[s], sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], ql[s], sacc[1], 0, 0, 0);
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], qh[s], sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], qh[s], sacc[1], 0, 0, 0);
    };
    f16x8 fr[2][4];                                                 // fragment double buffer
    // PP_SPREAD (DH = 64): the operands of k-step i+1 are fetched one fragment at a time BETWEEN the MFMAs of step i (read order 2, 3, 0, 1 =
    // the order in which the MFMAs of a step first touch them) instead of as a block in front of them
    constexpr bool SPREAD = PP_SPREAD && DT == 2;
    constexpr int XSM = SPREAD ? PP_XSM : 0;      // probability quarters 4 - XSM .. 3 (keys 16 q .. 16 q + 15 of the tile) are finished inside the next matrix phase
    static_assert(XSM == 0 || (PP_INIT_IN_ACC && !PP_PKADD && PP_P1 == 0), "deferred quarters exponentiate the accumulators as they are");
    static_assert(XSM >= 0 && XSM <= 2, "at most the two quarters of the second key block");
    // the deferred work of quarter q as 8 micro-steps (pair k = 0..3 of the lane's 8 logits: even step = 2 exp2 + 2 adds, odd step = the hi / lo split
    // of the pair, 3 instructions), dealt to the gaps between the MFMAs of P.V steps 0..2; step 3 (and step 2 when XSM = 2) consumes the result
    float xs_a = 0.f, xs_b = 0.f, xs_sum = 0.f;
    u32x4 xs_h[2], xs_l[2];
    auto xs_step = [&](int q, int i) __attribute__((always_inline)) {             // q = 2 or 3, i = 0..7
        const int k = i >> 1;
        if ((i & 1) == 0) {
            xs_a = fast_exp2(sacc[1][8 * (q & 1) + 2 * k]);
            xs_b = fast_exp2(sacc[1][8 * (q & 1) + 2 * k + 1]);
            xs_sum += xs_a;
            xs_sum += xs_b;
        } else {
            unsigned hi, lo;
            imp_split2(xs_a, xs_b, hi, lo);
            xs_h[q & 1][k] = hi; xs_l[q & 1][k] = lo;
            if (k == 3) {
                ph[1][q & 1] = __builtin_bit_cast(f16x8, xs_h[q & 1]);
                pl[1][q & 1] = __builtin_bit_cast(f16x8, xs_l[q & 1]);
            }
        }
    };
    // slot = index of the MFMA (0..17) of P.V steps 0..2 behind which the micro-step goes
    auto xs_slot = [&](int slot) __attribute__((always_inline)) {
        if (XSM == 1) {                            // quarter 3: every second gap of steps 0..2 (8 of 9)
            if ((slot & 1) == 1 && slot / 2 < 8) xs_step(3, slot / 2);
        } else if (XSM == 2) {                     // quarter 2 behind MFMAs 0..7 (ready for step 2 = MFMA 12), quarter 3 behind MFMAs 8..15
            if (slot < 8) xs_step(2, slot);
            else if (slot < 16) xs_step(3, slot - 8);
        }
    };
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
                auto xs = [&](int m) { if (XSM > 0 && g < 3) { xs_slot(6 * g + m); PP_SB(); } };
                PP_SB();
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], ph[jb][s2], oacc[0], 0, 0, 0); PP_SB();
                rd(2); PP_SB(); xs(0);
                oacc[DT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], ph[jb][s2], oacc[DT - 1], 0, 0, 0); PP_SB();
                rd(3); PP_SB(); xs(1);
                if (!PP_P1) { oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], pl[jb][s2], oacc[0], 0, 0, 0); PP_SB(); }
                rd(0); PP_SB(); xs(2);
                if (!PP_P1) { oacc[DT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], pl[jb][s2], oacc[DT - 1], 0, 0, 0); PP_SB(); }
                rd(1); PP_SB(); xs(3);
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], ph[jb][s2], oacc[0], 0, 0, 0); PP_SB(); xs(4);
                oacc[DT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], ph[jb][s2], oacc[DT - 1], 0, 0, 0); PP_SB(); xs(5);
                if (XSM > 0 && g == 2) { l_run += xs_sum; xs_sum = 0.f; }
            } else {
                if (g < 3) read_v(slot, g + 1, fr[(g + 1) & 1]);
                else if (kslot >= 0) read_k(kslot, 0, fr[0]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_v(g, fr[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto qk_mfmas = [&](int kslot, bool prefetched, int dma_spread = 0, int dma_t = 0) __attribute__((always_inline)) {     // dma_spread (a scalar 0 / 1): the requests of tile dma_t between the k-steps
        if (!prefetched) read_k(kslot, 0, fr[0]);
#if !(PP_CNEG && PP_INIT_IN_ACC)
        const float c0 = PP_INIT_IN_ACC ? -m_ref : 0.f;         // (0: the compiler feeds the first MFMA of each chain an inline zero)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[jb][r] = c0;
#endif
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (SPREAD) {
                f16x8 (&f)[4] = fr[s & 1];
                f16x8 (&n)[4] = fr[(s + 1) & 1];
                auto rd = [&](int i) { if (s + 1 < KS) read_k1(kslot, s + 1, i, n[i]); };
                constexpr bool CN = PP_CNEG && PP_INIT_IN_ACC;
                PP_SB();
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2], qh[s], (CN && s == 0) ? cneg : sacc[0], 0, 0, 0); PP_SB();
                rd(2); PP_SB();
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[3], qh[s], (CN && s == 0) ? cneg : sacc[1], 0, 0, 0); PP_SB();
                rd(3); PP_SB();
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], ql[s], sacc[0], 0, 0, 0); PP_SB();
                rd(0); PP_SB();
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], ql[s], sacc[1], 0, 0, 0); PP_SB();
                rd(1); PP_SB();
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], qh[s], sacc[0], 0, 0, 0); PP_SB();
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], qh[s], sacc[1], 0, 0, 0); PP_SB();
                if (DMA && PP_DMA_SPREAD && dma_spread != 0) {      // this wave's requests of tile dma_t: one behind every k-step, the fifth behind the last
                    static_assert(!DMA || DMA_NP == KS + 1, "one piece per k-step and one more");
                    if (s == 0) dma_open(dma_t); else dma_piece(s);
                    if (s == KS - 1) dma_close();
                    PP_SB();
                }
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
                if (2 * jb + s2 >= 4 - XSM) continue;
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = fast_exp2(sacc[jb][8 * s2 + e] - delta);
#if PP_P1 == 2
                {   // single-half probabilities whose row sum is taken from the ROUNDED values (v_dot2_f32_f16 against {1, 1}: two halves per instruction,
                    // exact products, fp32 accumulation): the output stays a convex combination of the value rows, and the fp32 adds disappear
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    u32x4 hw;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        h2 h; h[0] = (_Float16)pv[e]; h[1] = (_Float16)pv[e + 1];
                        ls2[0] = __builtin_amdgcn_fdot2(h, h2{(_Float16)1.f, (_Float16)1.f}, ls2[0], false);
                        hw[e >> 1] = __builtin_bit_cast(unsigned, h);
                    }
                    ph[jb][s2] = __builtin_bit_cast(f16x8, hw);
                    continue;
                }
#endif
#if PP_PKADD
#pragma unroll
                for (int e = 0; e < 8; e += 2) ls2 += f32x2{pv[e], pv[e + 1]};
#else
#pragma unroll
                for (int e = 0; e < 8; ++e) ls2[0] += pv[e];
#endif
#if PP_WHATIF & 2
                { u32x4 hw; for (int e = 0; e < 8; e += 2) { imp_f16x2 h; h[0] = (_Float16)pv[e]; h[1] = (_Float16)pv[e + 1]; hw[e >> 1] = __builtin_bit_cast(unsigned, h); }
                  ph[jb][s2] = pl[jb][s2] = __builtin_bit_cast(f16x8, hw); }
#else
                split8(pv, ph[jb][s2], pl[jb][s2]);
#endif
            }
        return ls2[0] + ls2[1];
    };

#ifdef PP_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
    const unsigned long long t_loop = tlast;
#endif
    // a counted wait assumes that nothing but the wave's own DMA pieces entered its VMEM queue behind the tile it waits for: wave 0 of a masked launch
    // (mask bytes) and the timeline build (stamp stores inside the loop) wait for everything instead
#if defined(PP_TIMELINE)
    const bool dma_counted = false;
#else
    const bool dma_counted = !(MASKED && wave == 0);
#endif
    auto tile_step = [&](int t, f32x4 (&rk)[LK], f32x4 (&rv)[LK], unsigned char& rb, f32x4 (&rk2)[LK], f32x4 (&rv2)[LK], unsigned char& rb2) __attribute__((always_inline)) {
        PP_CLK(7);
#ifdef PP_TIMELINE
        const bool tl_on = blockIdx.x == 0 && t >= PP_TL_T0 && t < PP_TL_T0 + 4;
#endif
        PP_TL(0);
        // =============================== X(t): matrix phase ===============================================
#if PP_PRIO == 0
        __builtin_amdgcn_s_setprio(1);
#endif
#if PP_LOADS_IN_X
        if (PP_STRAIGHT || t + 3 < nt) load_tile(t + 3, rk2, rv2, rb2);          // the other register set was converted in Y(t-1)
#endif
        if (DMA && !PP_DMA_SPREAD && group == 0 && t + 2 < dnt) dma_issue(t + 2);     // barrier phase 2t = 2 (t + 2) - 4
        if (t > 0) pv_mfmas((t - 1) & 3, t & 3, PP_PREFETCH != 0);
        // (PP_DMA_SPREAD: tile t + 2 in barrier phase 2t / 2t + 1: 2 (t + 2) - 4 at the earliest.  The flag as an opaque scalar int: ONE copy of the MFMA stream - two call sites cost
        // 8 register-pair copies of the prefetched fragments per phase - and a scalar compare + branch per request - a bool travels as a lane mask and is inverted on the VALU)
        int dma_sp = 0;
        if (DMA && PP_DMA_SPREAD && t + 2 < dnt) {
            dma_sp = ((dnk - (dt0 + t + 3) * KT) >> 31) + 1;                      // 1: all 64 keys of tile t + 2 exist (integer arithmetic: a bool -> int conversion goes through the VALU)
            if (dma_sp == 0) dma_issue(t + 2);                                    // a partial last tile: its five requests at once, rows clamped
        }
        qk_mfmas(t & 3, t > 0, dma_sp, t + 2);
#if PP_PRIO == 0
        __builtin_amdgcn_s_setprio(0);
#endif
        PP_CLK(0);
        PP_TL(1);
        if (DMA && group == 1) dma_wait(dma_counted && t + 2 < dnt);             // tile t + 1 before the barrier that ends phase 2t + 1 (behind it: the pieces of t + 2, requested in Y(t - 1) or in this X(t))
        PP_BARRIER();
        PP_CLK(1);
        PP_TL(2);
        // =============================== Y(t): vector phase ===============================================
        if (DMA && !PP_DMA_SPREAD && group == 1 && t + 3 < dnt) dma_issue(t + 3);     // barrier phase 2t + 2 = 2 (t + 3) - 4
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
#if PP_INIT_IN_ACC
        const float base = 0.f;                 // the accumulators already hold S - m_ref
#else
        const float base = m_ref;               // subtracted here: 32 v_sub in the vector phase instead of 32 v_mov in the matrix phase
#endif
        if (!slow) {
            lsum = probabilities(base);
            bool big = !(lsum < P_SUM_LIMIT);                     // also catches inf / nan
            if (XSM > 0) {                                        // the deferred quarters have no sum yet: bound their logits instead (2^14 each at most)
                float dmax = -INFINITY;
#pragma unroll
                for (int r = 16 - 8 * XSM; r < 16; ++r) dmax = fmaxf(dmax, sacc[1][r]);
                big = big || !(dmax - base < 14.f);
            }
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
#if PP_CNEG && PP_INIT_IN_ACC
#pragma unroll
            for (int r = 0; r < 16; ++r) cneg[r] = -m_ref;
#endif
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            lsum = probabilities(base + delta);
            if (XSM > 0) {                                        // the matrix phase will exponentiate these as they are: move them to the new reference here
#pragma unroll
                for (int r = 16 - 8 * XSM; r < 16; ++r) sacc[1][r] -= base + delta;
            }
            need_slow = __any(tmax == -INFINITY && !(lq > 0.f));
        }
        l_run += lsum;
        PP_CLK(2);
        PP_TL(3);
#if PP_PREFETCH
        read_v(t & 3, 0, fr[0]);                                  // operands of the first k-step of X(t+1) (or of the final P.V): V(t) has been in LDS since phase 2t-2
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !(PP_WHATIF & 1)
        if (t + 2 < nt) store_tile((t + 2) & 3, rk, rv, rb);      // this set holds tile t+2 (same parity as t)
#if !PP_LOADS_IN_X
        if (PP_STRAIGHT || t + 4 < nt) load_tile(t + 4, rk, rv, rb);
#endif
#endif
        PP_CLK(3);
        PP_TL(4);
        if (DMA && group == 0) dma_wait(dma_counted && t + 2 < dnt);             // tile t + 1 before the barrier that ends phase 2t + 1 (behind it: the pieces of t + 2, requested in X(t))
        PP_BARRIER();
        PP_CLK(4);
        PP_TL(5);
    };
#if PP_STRAIGHT
    // (the odd tail tile outside the loop: a conditional second step inside it leaves a path on which set A is the YOUNGEST staged tile at the
    // loop header, and the compiler then waits for every outstanding load - vmcnt(3..0) - before the stores of the first step)
    for (int t = 0; t + 1 < nt; t += 2) {
        tile_step(t, rkA, rvA, rbA, rkB, rvB, rbB);
        tile_step(t + 1, rkB, rvB, rbB, rkA, rvA, rbA);
    }
    if (nt & 1) tile_step(nt - 1, rkA, rvA, rbA, rkB, rvB, rbB);
#else
    for (int t = 0; t < nt; t += 2) {
        tile_step(t, rkA, rvA, rbA, rkB, rvB, rbB);
        if (t + 1 < nt) tile_step(t + 1, rkB, rvB, rbB, rkA, rvA, rbA);
    }
#endif
    pv_mfmas((nt - 1) & 3, -1, PP_PREFETCH != 0);
#ifdef PP_TIMELINE
    if (blockIdx.x == 0 && lane == 0) pp_hwid[wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: wave [3:0], simd [5:4], cu [11:8], sh, se
#endif
#ifdef PP_PROFILE
    const unsigned long long t_loop_end = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) pp_prof[wave][i] = prof[i];
#endif
    if (group == 0) PP_BARRIER();               // balance the extra barrier of waves 4-7
    PP_BARRIER();                               // everyone is done with the ring: reuse it for the transposition

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    constexpr int LDO = DH + 1;
    float* ot = smem + wave * 32 * LDO;
    if (ns > 1) {
        // ---- key split: this workgroup holds a PARTIAL result (O relative to its own m_ref, l, m_ref).  It goes to scratch
        // with agent-coherent write-through stores; a ticket per (pair, side, head, query tile) tells the LAST of the nsplit
        // workgroups to arrive, and that one merges: m = max m_s, O = sum_s 2^(m_s - m) O_s, l likewise, out = O / l.
        // Nobody waits for anybody (no spin): the ticket is a relaxed agent-scope atomic taken after the stores are acknowledged.
        constexpr int PART = 256 * DH + 512;                  // floats per partial: O [256][DH] | m [256] | l [256]
        const long unit = (((long)b * p.nside + sidx) * IMP_NUM_HEADS + h) * qtiles + qt;
        float* wsu = p.split_ws + unit * nsplit * PART;
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wsu, 0, (unsigned)(nsplit * PART * 4), 0x00020000);
        // rows staged with a 16-byte aligned pitch: 128-bit write-through stores, LPR lanes per row, RPI rows per instruction
        constexpr int LDP = DH + 4, LPR = DH / 4, RPI = 64 / LPR;
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
                                                   rsW, ((sp * PART) + (wave * 32 + qi) * DH + pc4) * 4, 0, 16);
        }
        if (half == 0) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m_ref), rsW, (sp * PART + 256 * DH + wave * 32 + l31) * 4, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l_tot), rsW, (sp * PART + 256 * DH + 256 + wave * 32 + l31) * 4, 0, 16);
        }
        __builtin_amdgcn_s_waitcnt(0);                         // the write-through stores are acknowledged
        __syncthreads();
        __shared__ int s_last;
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.split_cnt + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)ns - 1;
            if (s_last) __hip_atomic_store(p.split_cnt + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
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
        constexpr int LDP = DH + 4, LPR = DH / 4, RPI = 64 / LPR;
        float* otp = smem + wave * 32 * LDP;
        const float inv_l = 1.0f / l_tot;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                float q;
                if (PP_RCP) q = oacc[d][r] * inv_l;
                else if (PP_DIV3) q = imp_div_by(oacc[d][r], l_tot, inv_l);
                else q = oacc[d][r] / l_tot;
                otp[l31 * LDP + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = q;
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
#ifdef PP_PROFILE
    __builtin_amdgcn_s_waitcnt(0);              // the stores of this wave are out
    if (blockIdx.x == 0 && lane == 0) {
        pp_span[wave][0] = t_loop - t_entry; pp_span[wave][1] = t_loop_end - t_loop; pp_span[wave][2] = __builtin_readcyclecounter() - t_loop_end;
    }
#endif
}

}  // namespace
int imp_attn_dma_override = -1;      // probes / tests: 0 | 1 forces the staging variant of the ping-pong kernel for the launches that follow (-1: IMP_ATTN_DMA or the default)
namespace {

template <int DH>
hipError_t launch_pp(const AttnParams& p, int batch, int maxq, int nsplit, hipStream_t stream) {
    const int qtiles = (maxq + 255) / 256;
    const int total = qtiles * IMP_NUM_HEADS * p.nside * batch * nsplit;
    const size_t lds = (size_t)(4 * KT * (DH + 4) + 4 * KT * (DH + 16) + 4 * KT) * sizeof(float);
    const bool masked = p.side[0].kmask != nullptr || (p.nside == 2 && p.side[1].kmask != nullptr);
    // LDS-DMA staging of the ring (the kernel's DMA parameter): split-half K / V images at DH = 64; IMP_ATTN_DMA=0|1 overrides the default
    static const int dma_env = [] { const char* e = getenv("IMP_ATTN_DMA"); return e ? atoi(e) : PP_DMA_DEFAULT; }();
    if constexpr (DH == 64) {
        bool dma_ok = (imp_attn_dma_override >= 0 ? imp_attn_dma_override : dma_env) != 0 && p.kv_planes && (p.ldk & 3) == 0;
        for (int s = 0; s < p.nside; ++s)          // 16-byte sources: the images of whole head segments at 16-byte aligned rows
            dma_ok = dma_ok && ((reinterpret_cast<size_t>(p.side[s].k) | reinterpret_cast<size_t>(p.side[s].v) | (size_t)(p.side[s].sk_b * 4)) & 15) == 0;
        if (dma_ok) {
            if (masked) {
                if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_pp_kernel<DH, true, true>, lds)) return e;
                hipLaunchKernelGGL((attn_f16x3_pp_kernel<DH, true, true>), dim3(total), dim3(512), lds, stream, p, qtiles, total, nsplit);
            } else {
                if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_pp_kernel<DH, false, true>, lds)) return e;
                hipLaunchKernelGGL((attn_f16x3_pp_kernel<DH, false, true>), dim3(total), dim3(512), lds, stream, p, qtiles, total, nsplit);
            }
            return hipGetLastError();
        }
    }
    if (masked) {
        if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_pp_kernel<DH, true>, lds)) return e;
        hipLaunchKernelGGL((attn_f16x3_pp_kernel<DH, true>), dim3(total), dim3(512), lds, stream, p, qtiles, total, nsplit);
    } else {
        if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f16x3_pp_kernel<DH, false>, lds)) return e;
        hipLaunchKernelGGL((attn_f16x3_pp_kernel<DH, false>), dim3(total), dim3(512), lds, stream, p, qtiles, total, nsplit);
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

// EXPERIMENT support: k / v head segments -> [hi | lo] half images in place (one wave per (row, head); lane = channel, 2 per lane at dh = 32)
__global__ __launch_bounds__(256) void attn_kv_planes_kernel(float* base, long rows, int ld, int col0, int dh) {
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (unit >= rows * IMP_NUM_HEADS) return;
    const long row = unit / IMP_NUM_HEADS;
    const int h = (int)(unit % IMP_NUM_HEADS);
    float* seg = base + row * ld + col0 + h * dh;
    const float x = lane < dh ? seg[lane] : 0.f;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    _Float16* o = reinterpret_cast<_Float16*>(seg);
    if (lane < dh) { o[lane] = hi; o[dh + lane] = lo; }
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

hipError_t launch_attn_kv_planes(float* base, long rows, int ld, int col0, int dh, hipStream_t stream) {
    const long units = rows * IMP_NUM_HEADS;
    hipLaunchKernelGGL(attn_kv_planes_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, base, rows, ld, col0, dh);
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
    // IMP_KV_IMAGE=0; split-half K / V images always ran here) is the price of results that depend on the pair alone.
    const int nsplit = attention_f16x3_splits(p, batch);
    return p.dh == 64 ? launch_pp<64>(p, batch, maxq, nsplit, stream) : launch_pp<32>(p, batch, maxq, nsplit, stream);
}
