// Weight-fragment GEMMs for the 1x1 convolutions of a GNN layer (nets/layers.py:119-120 q|k|v, :145-149 / :210-218 the MLP):
//
//   C[M x N] = epilogue( prologue(A)[M x K] . W[N x K]^T ),   K = 256 or 512, N a multiple of 128, W static
//
// and, CHAIN, a second GEMM on the tile the first one just produced (the NEXT layer's projection of the updated descriptors):
//
//   X' = X + mlp.3(relu(norm(H)))            [64 rows x 256]  -> memory AND, as split halves, back into LDS
//   Q|K|V = X' . Wproj^T + b                 [64 rows x 768]  (or the value projection alone, 256 columns, for a sharing layer)
//
// Structure: the weights are split into f16 hi / lo halves and re-ordered into MFMA fragment order ONCE, when they are loaded
// (`wf_pack`), so a wave fetches a fragment as one coalesced 1-KB load from L2 and the weights never touch LDS; the activations
// of a 64-row tile are converted ONCE into hi / lo half planes covering the whole K extent and every column pass of the tile
// re-reads them from LDS.  Arithmetic: split-half f16x3, products lo.hi + hi.lo + hi.hi in that order, fp32 accumulate - the
// scheme of gemm_f32.hip (PREC = 1).
//
// Round 3: a workgroup is 8 waves = TWO groups of 4 (one wave of each group per SIMD).  A column pass (128 columns: 4 waves x
// 64 rows x 32 columns) belongs to one group and the groups take alternate passes, so while a wave of one group runs its
// epilogue (LDS transposition, bias / residual, 16-byte global stores - no MFMA) its SIMD partner of the other group streams
// the MFMAs of ITS pass: the 4-wave kernel of round 2 left the matrix pipe idle for a third of a workgroup's life (K loop 3.3 k
// cycles, epilogue 1.8 k per pass, one after the other; interleaving the stores into the same wave's K loop doubled its time -
// stores and weight loads share one in-order counter).  Group 0 runs its K loops at raised priority, which keeps the two groups
// out of phase.  Staging is shared by all 512 threads.
//
// LDS plane: row pitch = 2 K + 16 bytes = 1 (mod 16) sixteen-byte slots, so the 16 rows of a ds_read_b128 lane group
// ({0-3,12-15,20-27} ...) fall into 16 distinct slots.  Epilogue: wave-private transposition of 16 rows x 32 columns at a time
// (row pitch 144 bytes): whatever the accumulator layout, the tile is read back row-major - lane = 4 consecutive columns of one
// of 8 rows - so bias and residual are coalesced 16-byte loads and every store instruction writes 8 FULL 128-byte lines.
//
// Two accumulator layouts: SWAP = 1 (weights as first MFMA operand: a register holds 4 consecutive columns of the lane's row)
// and SWAP = 0 (a lane holds one column of 32 rows: the per-block InstanceNorm statistics (sum, M2 about the block mean) of
// the MLP's first convolution are register reductions).  STATS launches finish with a ticket: the LAST workgroup of a
// (pair, image) to arrive merges the per-block statistics into (mean, rstd) - Chan's formula in fp64, the arithmetic and
// order of stats_finalize_kernel (gemm_f32.hip) - so the separate finalize launch of round 2 is gone.
#include "imp_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


namespace {

constexpr int WF_TM = 64;       // rows of a tile
constexpr int WF_TP = 144;      // row pitch of the transposition buffers (bytes)
constexpr int WF_TROWS = 16;    // rows per transposition
constexpr int WF_TBUF = WF_TROWS * WF_TP;   // bytes per wave
constexpr int AUX_SC1 = 16;     // agent-scope coherent access (write-through store / cache-bypassing load)

struct WfLane { int lane, half, w4, grp; };

// (rounds 4-5 measured non-temporal output stores and an image-per-XCD tile mapping as build switches: -1 % / -0.8 % on the pipeline; removed, logs in profiles/r05/envab_*.log)

#ifdef WF_PROFILE   // tools/build_variant.sh prof -DWF_PROFILE: cycle stamps of wave 0 of each group of every workgroup (staging, K loops, epilogues, total)
__device__ unsigned long long wf_prof[2048][2][4];
#define WF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define WF_T(x)
#endif

// K loop of one column pass: acc[i] (i = rows 32 i .. 32 i + 31 of the tile) += A . W^T over KK, A fragments from the LDS planes
// one k-step ahead, weight fragments four k-steps ahead (bh / bl hold the first four k-steps on entry and the first four of
// `wfollow` on exit)
template <int KK, int SWAP, int KP = KK>            // KP: K extent of the planes (row pitch 2 KP + 16); KK < KP: a K loop over part of the tile's K range
__device__ __forceinline__ void wf_kloop(f32x16 (&acc)[2], const unsigned char* planes, const int (&aoff)[2], const u32x4* wp, const u32x4* wfollow,
                                         u32x4 (&bh)[4], u32x4 (&bl)[4]) {
    constexpr int PLANE = WF_TM * (2 * KP + 16);
    constexpr int NS = KK / 16;
    f16x8 fah[2][2], fal[2][2];
    auto load_frag = [&](int st, int fb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fah[fb][i] = *reinterpret_cast<const f16x8*>(planes + aoff[i] + st * 32);
            fal[fb][i] = *reinterpret_cast<const f16x8*>(planes + PLANE + aoff[i] + st * 32);
        }
    };
    load_frag(0, 0);
    u32x4 nh[4], nl[4];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int c = st & 3;
        if (st + 1 < NS) load_frag(st + 1, (st + 1) & 1);
        if (c == 0) {
            const u32x4* wnext = st == NS - 4 ? wfollow : wp + 512;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { nh[cc] = wnext[cc * 128]; nl[cc] = wnext[cc * 128 + 64]; }
            wp = wnext;
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 wh = __builtin_bit_cast(f16x8, bh[c]), wl = __builtin_bit_cast(f16x8, bl[c]);
        if (SWAP) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fal[st & 1][i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fah[st & 1][i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fah[st & 1][i], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[st & 1][i], wh, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[st & 1][i], wl, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[st & 1][i], wh, acc[i], 0, 0, 0);
        }
        if (c == 3) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { bh[cc] = nh[cc]; bl[cc] = nl[cc]; }
        }
    }
}

// the residual of a wave's 64 x 32 output block in the read-back layout of the epilogue (requested before the K loop)
__device__ __forceinline__ void wf_load_residual(f32x4 (&rres)[2][4], const float* Rb, int ldr, int row0, int M, int cb, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = min(row0 + 32 * i + 8 * j + (lane >> 3), M - 1);
            rres[i][j] = *reinterpret_cast<const f32x4*>(Rb + (long)row * ldr + cb + (lane & 7) * 4);
        }
}

// Epilogue of one column pass of one wave: accumulators -> (+ bias, + residual) -> C rows; optionally the same values as
// split halves into the LDS planes of the chained GEMM (TOPLANES; `pl2` = plane base, column cb of a 256-wide tile).
// `kv_image` (wave-uniform): this pass's columns are written as the split-half image the attention kernel stages by plain copy - per
// 64-channel head segment [64 hi halves | 64 lo halves] in the bytes of its 64 floats (attention_f16x3.hip, AttnParams::kv_planes)
template <int SWAP, int TOPLANES>
__device__ __forceinline__ void wf_epilogue(const f32x16 (&acc)[2], unsigned char* tbuf, const float* bias, const f32x4 (&rres)[2][4], bool has_res,
                                            float* Cb, int ldc, int row0, int M, int cb, const WfLane& L, int dbg, unsigned char* pl2, bool kv_image = false) {
    constexpr int PITCH2 = 2 * 256 + 16, PLANE2 = WF_TM * PITCH2;
    const int lane = L.lane, half = L.half;
    const int c4 = (lane & 7) * 4;
    const f32x4 bias4 = (bias && !(dbg & 4)) ? *reinterpret_cast<const f32x4*>(bias + cb + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {                            // rows 32 i + 16 c .. + 15 of the tile
            if (SWAP) {                                          // register r = column 4 half + (r & 3) + 8 (r >> 2) of row lane % 32
                if (((lane >> 4) & 1) == c) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e];
                        *reinterpret_cast<f32x4*>(tbuf + (lane & 15) * WF_TP + (4 * half + 8 * g) * 4) = v;
                    }
                }
            } else {                                             // register r = row 4 half + (r & 3) + 8 (r >> 2) of column lane % 32
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)
                    *reinterpret_cast<float*>(tbuf + (4 * half + (rr & 3) + 8 * (rr >> 2)) * WF_TP + (lane & 31) * 4) = acc[i][8 * c + rr];
            }
            // lanes exchange values through the wave's LDS buffer: wave-scope release / acquire around it (no hardware cost: LDS
            // operations of a wave execute in order; without it the compiler may keep a lane's EARLIER read of the same address
            // when that lane itself wrote nothing in between - it did, when only half of the lanes wrote the chunk)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = 8 * j + (lane >> 3);
                const int trow = 32 * i + 16 * c + lr;           // row of the tile
                const int row = row0 + trow;
                f32x4 v = *reinterpret_cast<const f32x4*>(tbuf + lr * WF_TP + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bias4[e];
                if (has_res && !(dbg & 4)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rres[i][2 * c + j][e];
                }
                if (kv_image) {
                    if (row < M && !(dbg & 1)) {
                        u32x2 hi, lo;
                        unsigned a, d;
                        imp_split2(v[0], v[1], a, d); hi[0] = a; lo[0] = d;
                        imp_split2(v[2], v[3], a, d); hi[1] = a; lo[1] = d;
                        const int col = cb + c4;                                  // head segment col & ~63, channel col & 63
                        unsigned char* seg = reinterpret_cast<unsigned char*>(Cb + (long)row * ldc + (col & ~63)) + (col & 63) * 2;
                        *reinterpret_cast<u32x2*>(seg) = hi;
                        *reinterpret_cast<u32x2*>(seg + 128) = lo;
                    }
                } else if (row < M && (!(dbg & 1) || v[0] == 123.456f)) {
                    *reinterpret_cast<f32x4*>(Cb + (long)row * ldc + cb + c4) = v;
                }
                if (TOPLANES) {                                  // (rows past M repeat row M - 1: they feed only rows that are never stored)
                    u32x2 hi, lo;
                    unsigned a, d;
                    imp_split2(v[0], v[1], a, d); hi[0] = a; lo[0] = d;
                    imp_split2(v[2], v[3], a, d); hi[1] = a; lo[1] = d;
                    unsigned char* dst = pl2 + trow * PITCH2 + (cb + c4) * 2;
                    *reinterpret_cast<u32x2*>(dst) = hi;
                    *reinterpret_cast<u32x2*>(dst + PLANE2) = lo;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the reads above precede the next chunk's writes of other lanes
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// per-block InstanceNorm statistics of one column pass (SWAP = 0 layout): (sum, M2 about the block mean) of the 64-row block for
// the lane's column, on the values as stored (bias included); both halves of the wave return the same pair
__device__ __forceinline__ void wf_block_stats(const f32x16 (&acc)[2], const float* bias, int row0, int M, int cb, const WfLane& L, float& sum_out,
                                               float& m2_out) {
    const int lane = L.lane, half = L.half;
    const int col = cb + (lane & 31);
    const float bvs = bias ? bias[col] : 0.f;
    const int nvalid = min(WF_TM, M - row0);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + 32 * i + 4 * half + (r & 3) + 8 * (r >> 2);
            if (row < M) sum += acc[i][r] + bvs;
        }
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)nvalid;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + 32 * i + 4 * half + (r & 3) + 8 * (r >> 2);
            const float d = (acc[i][r] + bvs) - mean;
            if (row < M) m2 = fmaf(d, d, m2);
        }
    m2 += __shfl_xor(m2, 32);
    sum_out = sum;
    m2_out = m2;
}
// ... written through to memory (the finalizing workgroup may sit on another XCD).  Called ONCE per wave, after its last pass: a
// write-through store is acknowledged only by the memory side (microseconds), and a wave's loads and stores retire in order - a
// statistics store between two passes stalled the next pass's first wait for weights (measured: +10 us per launch)
__device__ __forceinline__ void wf_store_stats(float* out_stats, int b, int rtile, int Mpad, int N, int cb, const WfLane& L, float sum, float m2) {
    if (L.half == 0) {
        const int tiles_side = (Mpad + WF_TM - 1) / WF_TM;         // the buffer is [b][this side's blocks (of the padded size)][N][2]
        float* o = out_stats + (((long)b * tiles_side + rtile) * N + cb) * 2;      // wave-uniform base, lane = column
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, 32u * 8u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(sum), __float_as_uint(m2)}, rs, (L.lane & 31) * 8, 0, AUX_SC1);
    }
}

// merged block statistics of one channel -> (mean, rstd): s1 = sum of the block sums, m2w = sum of the blocks' M2, sqn = sum of
// (block sum)^2 / (block rows) - Chan's parallel-variance formula; biased variance, eps inside the root (nets/layers.py:67-68).
// ONE function for the last-arriver merge of gemm_wf_kernel and for the slice owners of gemm_wf_fused_kernel: the two paths
// must produce the same bits
__device__ __forceinline__ float2 wf_finish_stats(double s1, double m2w, double sqn, int M, float eps) {
    const double mean = s1 / (double)M;
    double m2 = m2w + (sqn - (double)M * mean * mean);
    m2 = m2 < 0.0 ? 0.0 : m2;
    float2 o;
    o.x = (float)mean;
    o.y = (float)(1.0 / sqrt(m2 / (double)M + (double)eps));
    return o;
}

template <int K, int PRO, int SWAP, int STATS, int CHAIN>
__global__ __launch_bounds__(512) void gemm_wf_kernel(const WfParams p, int row_tiles) {
    constexpr int PITCH = 2 * K + 16;               // bytes per row of one half plane
    constexpr int PLANE = WF_TM * PITCH;
    constexpr int NS = K / 16;                      // k-steps
    extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];
    const int tid = threadIdx.x;
    WfLane L;
    L.lane = tid & 63; L.half = L.lane >> 5; L.w4 = (tid >> 6) & 3; L.grp = tid >> 8;
    const int lane = L.lane, half = L.half, w4 = L.w4, grp = L.grp;
    int z = blockIdx.x;
    // small launches: the column passes of a tile are dealt to `psplit` workgroups (each stages the tile itself) to fill the chip
    const int psplit = p.pass_split > 1 ? p.pass_split : 1;
    const int pgrp = z % psplit; z /= psplit;
    const int rtile = z % row_tiles; z /= row_tiles;
    const int sidx = z % p.nside;
    const int b = z / p.nside;
    const WfSide& S = p.side[sidx];
    const int Mpad = S.M;                           // strides and the statistics layout follow the padded size
    const int M = imp_count(p.rc, S.img, b, Mpad), N = p.N;     // ragged batches: this pair's own row count (0 = retired pair)
    const int row0 = rtile * WF_TM;
    if (row0 >= M) return;                          // uniform per workgroup, before any barrier
    WF_T(t_begin);
#ifdef WF_PROFILE
    unsigned long long t_k = 0, t_e = 0;
#endif

    unsigned char* const tbase = wf_smem + 2 * PLANE;                   // 8 wave-private transposition buffers
    // ---- PRO: the InstanceNorm constants (mean, rstd) of this batch element's K channels -> LDS (in the transposition area: it
    // is not in use before the staging barrier)
    float* stl = reinterpret_cast<float*>(tbase);                       // [K][2]
    if (PRO) {
        const float* st = S.in_stats + (long)b * K * 2;
        for (int i = tid; i < 2 * K; i += 512) stl[i] = st[i];
        __syncthreads();
    }
    // ---- stage the tile: 64 rows x K fp32 -> hi / lo half planes (rows past M are clamped: they only feed rows that are never stored)
    {
        const float* A = S.A + b * S.sA_b;
        const float* A2 = S.A2 ? S.A2 + b * S.sA2_b : A;
        constexpr int F4_ROW = K / 4;               // float4 per row
        constexpr int PER = WF_TM * F4_ROW / 512;   // float4 per thread
        f32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = j * 512 + tid;
            const int r = e / F4_ROW, k = (e - r * F4_ROW) * 4;
            const int gr = min(row0 + r, M - 1);
            const float* src = k < p.ksplit ? A + (long)gr * p.lda + k : A2 + (long)gr * p.lda2 + (k - p.ksplit);
            v[j] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = j * 512 + tid;
            const int r = e / F4_ROW, k = (e - r * F4_ROW) * 4;
            f32x4 x = v[j];
            if (PRO) {                              // InstanceNorm with the producer's finalised statistics + ReLU (nets/layers.py:67-76)
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(stl + 2 * k), s1 = *reinterpret_cast<const f32x4*>(stl + 2 * k + 4);
                x[0] = fmaxf((x[0] - s0[0]) * s0[1], 0.f);
                x[1] = fmaxf((x[1] - s0[2]) * s0[3], 0.f);
                x[2] = fmaxf((x[2] - s1[0]) * s1[1], 0.f);
                x[3] = fmaxf((x[3] - s1[2]) * s1[3], 0.f);
            }
            u32x2 hi, lo;
            unsigned a, c;
            imp_split2(x[0], x[1], a, c); hi[0] = a; lo[0] = c;
            imp_split2(x[2], x[3], a, c); hi[1] = a; lo[1] = c;
            unsigned char* dst = wf_smem + r * PITCH + k * 2;
            *reinterpret_cast<u32x2*>(dst) = hi;
            *reinterpret_cast<u32x2*>(dst + PLANE) = lo;
        }
    }
    __syncthreads();
    WF_T(t_staged);

    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = (32 * i + (lane & 31)) * PITCH + half * 16;
    const int npass_all = N >> 7;
    const int npass = npass_all / psplit;          // passes of this workgroup: pbase .. pbase + npass, dealt alternately to the two groups
    const int pbase = pgrp * npass;
    const u32x4* wbase = reinterpret_cast<const u32x4*>(p.Wf_) + lane;
    auto wptr = [&](int pass) { return wbase + (size_t)(pass * 4 + w4) * NS * 128; };
    unsigned char* const tbuf = tbase + (grp * 4 + w4) * WF_TBUF;
    float* const Cb = S.C + b * S.sC_b;
    const float* const Rb = S.R ? S.R + b * S.sR_b : nullptr;
    // workgroups start at different column passes and wrap around: they all begin at the same time, and walking the weights in
    // the same order would have every CU ask the L2 for the same lines at the same moment
    const int mine = (npass + 1 - grp) >> 1;       // passes of this group: local indices grp, grp + 2, ...
    const int rot = mine > 0 ? (int)((blockIdx.x / psplit) % (unsigned)mine) : 0;
    auto pass_of = [&](int t) { int u = t + rot; if (u >= mine) u -= mine; return pbase + grp + 2 * u; };

    if (CHAIN) {
        // ================= first GEMM (N = 256: one pass per group), epilogue behind a barrier, then the chained projection =================
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int pass = grp;
        const int cb = pass * 128 + w4 * 32;
        f32x4 rres[2][4];
        if (Rb && !(p.dbg & 4)) wf_load_residual(rres, Rb, p.ldr, row0, M, cb, lane);
        u32x4 bh[4], bl[4];
        {
            const u32x4* w0 = wptr(pass);
#pragma unroll
            for (int c = 0; c < 4; ++c) { bh[c] = w0[c * 128]; bl[c] = w0[c * 128 + 64]; }
        }
        constexpr int NS2 = 256 / 16;
        const int npass2 = p.N2 >> 7;
        const int mine2 = (npass2 + 1 - grp) >> 1;
        const u32x4* wbase2 = reinterpret_cast<const u32x4*>(p.Wf2_) + lane;
        auto wptr2 = [&](int ps) { return wbase2 + (size_t)(ps * 4 + w4) * NS2 * 128; };
        wf_kloop<K, SWAP>(acc, wf_smem, aoff, wptr(pass), wptr2(grp), bh, bl);      // (bh / bl leave with the first fragments of the chained GEMM)
        __syncthreads();                            // every wave is done with the K-wide planes: the new tile may overwrite them
        wf_epilogue<SWAP, 1>(acc, tbuf, p.bias, rres, Rb != nullptr, Cb, p.ldc, row0, M, cb, L, p.dbg, wf_smem);
        __syncthreads();                            // the 64 x 256 tile X' is complete in LDS (pitch 528 bytes, planes 64 x 528 apart)
        constexpr int PITCH2 = 2 * 256 + 16;
        int aoff2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) aoff2[i] = (32 * i + (lane & 31)) * PITCH2 + half * 16;
        float* const C2 = S.C2 + b * S.sC2_b;
        const f32x4 (&nores)[2][4] = rres;
        if (grp == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int t = 0; t < mine2; ++t) {
            const int ps = grp + 2 * t;
            const int pn = t + 1 < mine2 ? ps + 2 : ps;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            wf_kloop<256, SWAP>(acc, wf_smem, aoff2, wptr2(ps), wptr2(pn), bh, bl);
            wf_epilogue<SWAP, 0>(acc, tbuf, p.bias2, nores, false, C2, p.ldc2, row0, M, ps * 128 + w4 * 32, L, p.dbg, nullptr, ps * 128 >= p.kv_image_col2);
        }
        return;
    }

    u32x4 bh[4], bl[4];
    if (mine > 0) {
        const u32x4* w0 = wptr(pass_of(0));
#pragma unroll
        for (int c = 0; c < 4; ++c) { bh[c] = w0[c * 128]; bl[c] = w0[c * 128 + 64]; }
    }
    if (grp == 0) __builtin_amdgcn_s_setprio(1);
    float st_sum[4], st_m2[4];                      // STATS: (sum, M2) of this wave's passes, stored after the last one (N <= 1024: 4 passes per group)
    int st_cb[4];
#pragma unroll 1
    for (int t = 0; t < mine; ++t) {
        const int pass = pass_of(t);
        const int pnext = pass_of(t + 1 < mine ? t + 1 : 0);      // (after the last: a harmless load)
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int cb = pass * 128 + w4 * 32;                     // first column of this wave in this pass
        // the residual of this pass in the read-back layout of the epilogue, requested now: it arrives under the K loop
        f32x4 rres[2][4];
        if (Rb && !(p.dbg & 4)) wf_load_residual(rres, Rb, p.ldr, row0, M, cb, lane);
        WF_T(t0);
        wf_kloop<K, SWAP>(acc, wf_smem, aoff, wptr(pass), wptr(pnext), bh, bl);
        WF_T(t1);
        if (STATS) {                                               // (uniform selects instead of dynamic indexing: the arrays stay in registers)
            float su, m2v;
            wf_block_stats(acc, p.bias, row0, M, cb, L, su, m2v);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (t == q) { st_sum[q] = su; st_m2[q] = m2v; st_cb[q] = cb; }
        }
        if (p.dbg & 2) { if (acc[0][0] == 123.456f) Cb[0] = acc[1][5]; continue; }
        if (STATS && t == mine - 1) {
            // the write-through statistics stores of this wave leave before its LAST epilogue (no load follows them any more, so they
            // stall nothing): their acknowledgement - which the ticket below has to wait for - travels under the epilogue's stores
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < mine) wf_store_stats(S.out_stats, b, rtile, Mpad, N, st_cb[q], L, st_sum[q], st_m2[q]);
        }
        wf_epilogue<SWAP, 0>(acc, tbuf, p.bias, rres, Rb != nullptr, Cb, p.ldc, row0, M, cb, L, p.dbg, nullptr, SWAP && pass * 128 >= p.kv_image_col);
#ifdef WF_PROFILE
        { __builtin_amdgcn_s_waitcnt(0); WF_T(t2); t_k += t1 - t0; t_e += t2 - t1; }
#endif
    }
#ifdef WF_PROFILE
    if (w4 == 0 && lane == 0 && blockIdx.x < 2048) {
        wf_prof[blockIdx.x][grp][0] = t_staged - t_begin; wf_prof[blockIdx.x][grp][1] = t_k; wf_prof[blockIdx.x][grp][2] = t_e;
        wf_prof[blockIdx.x][grp][3] = __builtin_readcyclecounter() - t_begin;
    }
#endif
    if (STATS) {
        // ---- ticket: the last workgroup of this (pair, image) to get here turns the per-block statistics into (mean, rstd)
        if (!S.fin_stats) return;
        __builtin_amdgcn_s_waitcnt(0);              // this wave's write-through stores are acknowledged
        __syncthreads();
        __shared__ int s_last;
        const int tiles_side = (M + WF_TM - 1) / WF_TM;
        if (tid == 0) {
            // one ticket per (pair, image, pass group): with the column passes of a tile dealt to several workgroups (small launches) every
            // pass group finishes the statistics of ITS columns - the merge of a launch is then psplit last-arrivers working side by side
            // instead of one (the serial tail of a 32-tile launch was 5-6 us of 15)
            unsigned* cnt = p.stat_cnt + (b * p.nside + sidx) * WF_MAX_PSPLIT + pgrp;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)tiles_side - 1u;
            if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-armed for the next launch
        }
        __syncthreads();
        if (!s_last) return;
        // Chan's parallel-variance merge in fp64 in the arithmetic and order of stats_finalize_kernel: 4 block groups (group g takes
        // blocks g, g + 4, ...), combined ((g0 + g1) + (g2 + g3)) through LDS.  Thread = (group, column mod 128); all loads of a
        // round (up to 8 blocks x the pass group's columns / 128) are issued before any is used - they are cache-bypassing and ~2 us each
        const float* part = S.out_stats + (long)b * ((Mpad + WF_TM - 1) / WF_TM) * N * 2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, (unsigned)((size_t)tiles_side * N * 8), 0x00020000);
        double* sm = reinterpret_cast<double*>(wf_smem);              // [4 groups][3][N]: the planes are no longer needed
        const int g = tid >> 7, kk = tid & 127;
        constexpr int NR = 4;                                          // column rounds held in flight (N <= 512)
        const int nr = npass;                                          // this pass group's columns: [c0, c0 + 128 nr)
        const int c0 = pbase * 128;
        double a1[NR], a2[NR], a3[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) a1[r] = a2[r] = a3[r] = 0.0;
        for (int t0 = g; t0 < tiles_side; t0 += 32) {
            u32x2 raw[NR][8];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + 4 * u;
                    raw[r][u] = (r < nr && t < tiles_side) ? __builtin_amdgcn_raw_buffer_load_b64(rs, (t * N + c0 + kk + 128 * r) * 8, 0, AUX_SC1) : u32x2{0u, 0u};
                }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 4 * u;
                if (t < tiles_side) {
                    const int nt = min(WF_TM, M - t * WF_TM);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const double vx = (double)__uint_as_float(raw[r][u][0]), vy = (double)__uint_as_float(raw[r][u][1]);
                        a1[r] += vx; a2[r] += vy; a3[r] += vx * vx / (double)nt;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r < nr) {
                const int k = kk + 128 * r;
                sm[(g * 3 + 0) * N + k] = a1[r]; sm[(g * 3 + 1) * N + k] = a2[r]; sm[(g * 3 + 2) * N + k] = a3[r];
            }
        __syncthreads();
        for (int k = tid; k < 128 * nr; k += 512) {
            const double s1t = (sm[(0 * 3 + 0) * N + k] + sm[(1 * 3 + 0) * N + k]) + (sm[(2 * 3 + 0) * N + k] + sm[(3 * 3 + 0) * N + k]);
            const double m2w = (sm[(0 * 3 + 1) * N + k] + sm[(1 * 3 + 1) * N + k]) + (sm[(2 * 3 + 1) * N + k] + sm[(3 * 3 + 1) * N + k]);
            const double sqn = (sm[(0 * 3 + 2) * N + k] + sm[(1 * 3 + 2) * N + k]) + (sm[(2 * 3 + 2) * N + k] + sm[(3 * 3 + 2) * N + k]);
            reinterpret_cast<float2*>(S.fin_stats)[(long)b * N + c0 + k] = wf_finish_stats(s1t, m2w, sqn, M, p.norm_eps);
        }
    }
}

// ====================================================================================================================
// FUSED layer MLP (round 4).  Two launches per layer used to frame the InstanceNorm between mlp.0 and mlp.3:
// MLP0 wrote the hidden tensor (33.5 MB at B = 4, N = 2048) and its statistics, MLP3 staged it again - a store drain, a launch
// boundary and a second staging phase with the matrix pipe idle, ~15 us per layer.  Here a workgroup keeps its 64 x 512 hidden
// tile in the accumulators of its two MLP0 column passes per group (64 VGPRs per lane) while the statistics travel:
//   1. stage [x | attention output] -> half planes, mlp.0 K loops (2 passes per group), per-block (sum, M2) as in gemm_wf_kernel
//   2. reduce-scatter: every workgroup publishes its 512 block records as granules {sum, tag, M2, tag}; tile j of the image OWNS
//      the channel slice [j S, (j + 1) S), S = ceil(512 / tiles): it gathers that slice from all blocks (tagged polls, no ticket,
//      no store acknowledgement), merges it with Chan's formula in fp64 in the order of the last-arriver merge above and publishes
//      {mean, tag, rstd, tag}
//   3. all-gather: every thread polls one channel's (mean, rstd) into LDS
//   4. (acc + bias - mean) * rstd, ReLU, hi / lo split in registers; neighbouring lanes trade one value (DPP) so that every lane
//      writes 2 consecutive channels of one row: the planes of the hidden tile replace the dead input tile in LDS
//   5. mlp.3 + bias + residual -> new descriptors (memory and planes), then the chained projection: the CHAIN branch of gemm_wf_kernel
// Same operands, same instruction sequences per value, same merge order: results are bit-identical to the two-launch path.
// Exchange = two one-hop granule sweeps (handoff rows of MI355X_MICROARCH.md): writers use write-through (sc1) 16-byte stores, readers
// poll with cache-bypassing loads until both tags of a granule pair match; tags are unique per launch on the buffers.  Polls are
// bounded: a time-out raises *status = 3 (seen by every waiter within 256 polls) and the mapped host word, the workgroup poisons
// its rows (NaN residual -> NaN descriptors -> NaN scores -> no matches), and the library reports IMP_E_RESIDENT at its next entry.
constexpr int WF_SPIN_LIMIT = 1 << 21;
constexpr int AUX_POLL = AUX_SC1 | (int)0x80000000;      // + volatile: a poll must stay inside its loop

// (a wait that ran out also leaves the post-mortem record of imp_kernels.h imp_postmortem_write: which tile waited for which record, with what tag)
struct WfWait { unsigned tag; int pair, tile, tiles, B; };
__device__ __forceinline__ void wf_poll_health(int* status, int* host, int spins, bool& dead, const WfWait& w, int phase, int idx, unsigned seen) {
    if (spins > WF_SPIN_LIMIT) {
        __hip_atomic_store(status, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host) {
            imp_postmortem_write(host, 3, w.tag, phase, idx, w.tag, seen, -1, w.pair, w.tile, w.tiles, 0, w.B);
            __hip_atomic_store(host, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
}

// normalise one held column pass (SWAP = 0 layout: lane = column cb + lane % 32, register r of acc[i] = row 32 i + 4 half + (r & 3)
// + 8 (r >> 2)) and write it into the K = 512 half planes as columns cb .. cb + 31 of the hidden tile.  Lanes l and l ^ 1 trade one
// value per register pair (r, r + 4): the even lane ends up with columns (k, k + 1) of row rho, the odd lane with the same two
// columns of row rho + 8 - 4-byte LDS writes of two halves; the 64 lanes of a write cover 4 rows x 16 dwords in distinct banks
// (row pitch 260 dwords: rows rho, rho + 8, rho + 4, rho + 12 start at banks 0, 32, 16, 48 relative to the first)
__device__ __forceinline__ void wf_hidden_to_planes(f32x16 (&acc)[2], float bv, const float* stl, unsigned char* planes, int cb, const WfLane& L) {
    constexpr int PITCH = 2 * 512 + 16, PLANE = WF_TM * PITCH;
    const int lane = L.lane, half = L.half;
    const int col = cb + (lane & 31);
    const float mean = stl[2 * col], rstd = stl[2 * col + 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(((acc[i][r] + bv) - mean) * rstd, 0.f);      // the stored value of MLP0 (acc + bias), then nets/layers.py:67-76 as the PRO staging does it
    const bool odd = lane & 1;
    unsigned char* const colbase = planes + (cb + (lane & 30)) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ra = (q & 3) + 8 * (q >> 2), rb = ra + 4;                  // rows rho and rho + 8
            float xa = acc[i][ra], xb = acc[i][rb];
            asm("" : "+v"(xa), "+v"(xb));           // (opaque copies: the compiler otherwise turns the selects below into a lane-varying INDEX into acc - a 16-way compare / select chain per value)
            const float send = odd ? xa : xb;
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]: lane ^ 1
            const float v0 = odd ? recv : xa;
            const float v1 = odd ? xb : recv;
            const int row = 32 * i + 4 * half + (ra & 3) + 8 * (ra >> 2) + (odd ? 8 : 0);
            unsigned hi, lo;
            imp_split2(v0, v1, hi, lo);
            *reinterpret_cast<unsigned*>(colbase + row * PITCH) = hi;
            *reinterpret_cast<unsigned*>(colbase + row * PITCH + PLANE) = lo;
        }
}

__global__ __launch_bounds__(512) void gemm_wf_fused_kernel(const WfParams p, const WfFused f, int row_tiles) {
    constexpr int K = 512, N = 512;
    constexpr int PITCH = 2 * K + 16;
    constexpr int PLANE = WF_TM * PITCH;
    constexpr int NS = K / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];
    __shared__ int s_dead;
    const int tid = threadIdx.x;
    WfLane L;
    L.lane = tid & 63; L.half = L.lane >> 5; L.w4 = (tid >> 6) & 3; L.grp = tid >> 8;
    const int lane = L.lane, half = L.half, w4 = L.w4, grp = L.grp;
    int z = blockIdx.x;
    const int rtile = z % row_tiles; z /= row_tiles;
    const int sidx = z % p.nside;
    const int b = z / p.nside;
    const WfSide& S = p.side[sidx];
    const int Mpad = S.M;                           // strides and the layout of the exchange buffers follow the padded size
    const int M = imp_count(p.rc, S.img, b, Mpad);  // ragged batches: this pair's own row count (0 = retired pair)
    const int row0 = rtile * WF_TM;
    if (row0 >= M) return;                          // uniform per workgroup, before any barrier or exchange
    const int T = (M + WF_TM - 1) / WF_TM;          // workgroups (= statistics blocks) of this (pair, image)
    const int Tpad = (Mpad + WF_TM - 1) / WF_TM;
#ifdef WF_PROFILE
    unsigned long long tp[7];
    tp[0] = __builtin_readcyclecounter();
#define WF_TP(i) tp[i] = __builtin_readcyclecounter()
#else
#define WF_TP(i)
#endif
    if (tid == 0) s_dead = 0;
    unsigned char* const tbase = wf_smem + 2 * PLANE;                   // 8 wave-private transposition buffers; before: (mean, rstd) [512][2]
    float* const stl = reinterpret_cast<float*>(tbase);
    // ---- 1. stage the tile [x | attention output]: 64 rows x 512 fp32 -> hi / lo half planes (rows past M repeat row M - 1).  All loads
    // are requested at once; the x half (k < 256) is converted and handed to the matrix pipe first - the first 16 k-steps of every wave's
    // first column pass run while the attention half is still arriving (all workgroups of the launch stage together: 33.5 MB at the fabric's limit)
    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = (32 * i + (lane & 31)) * PITCH + half * 16;
    const u32x4* const wbase = reinterpret_cast<const u32x4*>(p.Wf_) + lane;
    const u32x4* const wbase3 = reinterpret_cast<const u32x4*>(f.Wf3_) + lane;
    auto wptr = [&](int pass) { return wbase + (size_t)(pass * 4 + w4) * NS * 128; };
    auto wptr3 = [&](int pass) { return wbase3 + (size_t)(pass * 4 + w4) * NS * 128; };
    // mlp.0: four column passes of 128, two per group (group g: passes g and g + 2; workgroups start at alternating ones)
    const int rot = blockIdx.x & 1;
    const int passA = grp + 2 * rot, passB = grp + 2 * (1 - rot);
    const int cbA = passA * 128 + w4 * 32, cbB = passB * 128 + w4 * 32;
    const float bvA = p.bias ? p.bias[cbA + (lane & 31)] : 0.f, bvB = p.bias ? p.bias[cbB + (lane & 31)] : 0.f;     // (held: no global load between the exchange and the planes)
    u32x4 bh[4], bl[4];
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[i][r] = 0.f; accB[i][r] = 0.f; }
    {
        const float* A = S.A + b * S.sA_b;
        const float* A2 = S.A2 ? S.A2 + b * S.sA2_b : A;
        constexpr int KH = K / 2;                   // = ksplit for a layer (x | attention output), any ksplit % 4 == 0 works
        constexpr int F4_ROW = KH / 4;              // float4 per row of a half
        constexpr int PER = WF_TM * F4_ROW / 512;   // float4 per thread and half
        f32x4 v[2][PER];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int e = j * 512 + tid;
                const int r = e / F4_ROW, k = h * KH + (e - r * F4_ROW) * 4;
                const int gr = min(row0 + r, M - 1);
                const float* src = k < p.ksplit ? A + (long)gr * p.lda + k : A2 + (long)gr * p.lda2 + (k - p.ksplit);
                v[h][j] = *reinterpret_cast<const f32x4*>(src);
            }
        {
            const u32x4* w0 = wptr(passA);
#pragma unroll
            for (int c = 0; c < 4; ++c) { bh[c] = w0[c * 128]; bl[c] = w0[c * 128 + 64]; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int e = j * 512 + tid;
                const int r = e / F4_ROW, k = h * KH + (e - r * F4_ROW) * 4;
                u32x2 hi, lo;
                unsigned a, c;
                imp_split2(v[h][j][0], v[h][j][1], a, c); hi[0] = a; lo[0] = c;
                imp_split2(v[h][j][2], v[h][j][3], a, c); hi[1] = a; lo[1] = c;
                unsigned char* dst = wf_smem + r * PITCH + k * 2;
                *reinterpret_cast<u32x2*>(dst) = hi;
                *reinterpret_cast<u32x2*>(dst + PLANE) = lo;
            }
            __syncthreads();
            if (h == 0) {
                WF_TP(1);
                if (grp == 0) __builtin_amdgcn_s_setprio(1);
                // k-steps 0 .. 15 of the first pass (weights of k-steps 16 .. follow in the fragment stream)
                wf_kloop<KH, 0, K>(accA, wf_smem, aoff, wptr(passA), wptr(passA) + (KH / 16) * 128, bh, bl);
                if (grp == 0) __builtin_amdgcn_s_setprio(0);
            }
        }
    }
    if (grp == 0) __builtin_amdgcn_s_setprio(1);
    float sumA, m2A, sumB, m2B;
    {
        int aoff_h[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) aoff_h[i] = aoff[i] + (K / 2) * 2;       // k = 256 .. of the planes
        wf_kloop<K / 2, 0, K>(accA, wf_smem, aoff_h, wptr(passA) + (K / 32) * 128, wptr(passB), bh, bl);
    }
    wf_block_stats(accA, p.bias, row0, M, cbA, L, sumA, m2A);
    wf_kloop<K, 0>(accB, wf_smem, aoff, wptr(passB), wptr(passB), bh, bl);      // (after the last pass: a harmless reload - mlp.3's first fragments
    wf_block_stats(accB, p.bias, row0, M, cbB, L, sumB, m2B);                   //  are requested after the exchange: 32 registers less to hold across it)
    if (grp == 0) __builtin_amdgcn_s_setprio(0);
    WF_TP(2);
    float* const Cb = S.C + b * S.sC_b;
    const float* const Rb = S.R ? S.R + b * S.sR_b : nullptr;
    const int cb3 = grp * 128 + w4 * 32;

    // ---- 2. publish this block's records; the slice owner merges
    const unsigned tag = f.tag;
    bool dead = false;
    const WfWait wfw{tag, b, rtile, T, (int)gridDim.x};
    {
        float* rec_tile = f.rec[sidx] + (((long)b * Tpad + rtile) * N) * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rec_tile, 0, (unsigned)(N * 16), 0x00020000);
        if (half == 0 && !(f.fake && blockIdx.x == 0)) {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(sumA), tag, __float_as_uint(m2A), tag}, rs, (cbA + lane) * 16, 0, AUX_SC1);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(sumB), tag, __float_as_uint(m2B), tag}, rs, (cbB + lane) * 16, 0, AUX_SC1);
        }
    }
    const int SL = (N + T - 1) / T;                  // channels per owner
    {
        const float* rec_all = f.rec[sidx] + (long)b * Tpad * N * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rec_all, 0, (unsigned)((size_t)T * N * 16), 0x00020000);
        float* fin = f.fin[sidx] + (long)b * N * 4;
        const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc((void*)fin, 0, (unsigned)(N * 16), 0x00020000);
        const int g = tid & 3;                       // block group of the merge: blocks g, g + 4, ... (the order of the last-arriver merge)
        for (int my = tid >> 2; my < SL; my += 128) {
            const int chan = rtile * SL + my;
            if (chan >= N) break;                    // (uniform over the four lanes of a channel)
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int t0 = g; t0 < T; t0 += 32) {
                // all (up to) 8 records of this round in flight at once, re-read together until every one carries this launch's tag
                // (a record never changes again inside a launch, so re-reading a good one is harmless)
                u32x4 raw[8];
                int spins = 0;
                for (;;) {
                    asm volatile("" ::: "memory");
                    bool all = true;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = t0 + 4 * u;
                        raw[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (min(t, T - 1) * N + chan) * 16, 0, AUX_POLL);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) all = all && raw[u][1] == tag && raw[u][3] == tag;      // (blocks past the last re-read block T - 1)
                    if (all || dead) break;
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 255) == 0) wf_poll_health(f.status, f.host_status, spins << 2, dead, wfw, 1, t0 * N + chan, raw[0][1]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + 4 * u;
                    if (t < T) {
                        const int nt = min(WF_TM, M - t * WF_TM);
                        const double vx = (double)__uint_as_float(raw[u][0]), vy = (double)__uint_as_float(raw[u][2]);
                        a1 += vx; a2 += vy; a3 += nt == WF_TM ? vx * vx * (1.0 / WF_TM) : vx * vx / (double)nt;     // (x / 64 == x * 2^-6 exactly: the bits of the division, without it)
                    }
                }
            }
            // ((g0 + g1) + (g2 + g3)): the combination order of the last-arriver merge (additions commute: both lanes of a pair hold the same bits)
            a1 += __shfl_xor(a1, 1); a2 += __shfl_xor(a2, 1); a3 += __shfl_xor(a3, 1);
            a1 += __shfl_xor(a1, 2); a2 += __shfl_xor(a2, 2); a3 += __shfl_xor(a3, 2);
            if (g == 0) {
                const float2 o = wf_finish_stats(a1, a2, a3, M, p.norm_eps);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o.x), tag, __float_as_uint(o.y), tag}, rsf, chan * 16, 0, AUX_SC1);
            }
        }
        // ---- 3. every thread fetches one channel's (mean, rstd)
        {
            u32x4 r;
            int spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                r = __builtin_amdgcn_raw_buffer_load_b128(rsf, tid * 16, 0, AUX_POLL);
                if ((r[1] == tag && r[3] == tag) || dead) break;
                __builtin_amdgcn_s_sleep(4);
                if ((++spins & 255) == 0) wf_poll_health(f.status, f.host_status, spins, dead, wfw, 2, tid, r[1]);
            }
            stl[2 * tid] = __uint_as_float(r[0]);
            stl[2 * tid + 1] = __uint_as_float(r[2]);
        }
    }
    if (dead) s_dead = 1;
    __syncthreads();                                // (mean, rstd) complete; every wave is done with the input planes
    WF_TP(3);
    const bool dead_wg = s_dead != 0;
    // mlp.3's first weight fragments and the residual of this wave's mlp.3 block, requested now: they arrive under step 4
    {
        const u32x4* w0 = wptr3(grp);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bh[c] = w0[c * 128]; bl[c] = w0[c * 128 + 64]; }
    }
    f32x4 rres[2][4];
    if (Rb) wf_load_residual(rres, Rb, p.ldr, row0, M, cb3, lane);
    // ---- 4. the hidden tile, normalised, as half planes over the dead input tile
    wf_hidden_to_planes(accA, bvA, stl, wf_smem, cbA, L);
    wf_hidden_to_planes(accB, bvB, stl, wf_smem, cbB, L);
    if (dead_wg) {                                  // unfinished exchange: this tile's statistics are garbage - its rows must not pass as results
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) rres[i][j][e] = __builtin_nanf("");
    }
    __syncthreads();
    WF_TP(4);
    // ---- 5. mlp.3 (N = 256: one pass per group) + bias + residual, then the chained projection (the CHAIN branch of gemm_wf_kernel)
    unsigned char* const tbuf = tbase + (grp * 4 + w4) * WF_TBUF;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bool has_res = Rb != nullptr || dead_wg;
    if (!p.Wf2_) {
        wf_kloop<K, 1>(acc, wf_smem, aoff, wptr3(grp), wptr3(grp), bh, bl);
        wf_epilogue<1, 0>(acc, tbuf, f.bias3, rres, has_res, Cb, p.ldc, row0, M, cb3, L, 0, nullptr);
        WF_TP(5); WF_TP(6);
    } else {
        constexpr int NS2 = 256 / 16;
        const int npass2 = p.N2 >> 7;
        const int mine2 = (npass2 + 1 - grp) >> 1;
        const u32x4* wbase2 = reinterpret_cast<const u32x4*>(p.Wf2_) + lane;
        auto wptr2 = [&](int ps) { return wbase2 + (size_t)(ps * 4 + w4) * NS2 * 128; };
        wf_kloop<K, 1>(acc, wf_smem, aoff, wptr3(grp), wptr2(grp), bh, bl);
        __syncthreads();                            // every wave is done with the K-wide planes: the new tile may overwrite them
        wf_epilogue<1, 1>(acc, tbuf, f.bias3, rres, has_res, Cb, p.ldc, row0, M, cb3, L, 0, wf_smem);
        __syncthreads();                            // the 64 x 256 tile X' is complete in LDS (pitch 528 bytes)
        WF_TP(5);
        constexpr int PITCH2 = 2 * 256 + 16;
        int aoff2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) aoff2[i] = (32 * i + (lane & 31)) * PITCH2 + half * 16;
        float* const C2 = S.C2 + b * S.sC2_b;
        if (grp == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int t = 0; t < mine2; ++t) {
            const int ps = grp + 2 * t;
            const int pn = t + 1 < mine2 ? ps + 2 : ps;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            wf_kloop<256, 1>(acc, wf_smem, aoff2, wptr2(ps), wptr2(pn), bh, bl);
            wf_epilogue<1, 0>(acc, tbuf, p.bias2, rres, false, C2, p.ldc2, row0, M, ps * 128 + w4 * 32, L, 0, nullptr, ps * 128 >= p.kv_image_col2);
        }
        WF_TP(6);
    }
#ifdef WF_PROFILE
    if (f.prof && (tid & 255) == 0) {
        unsigned long long* o = f.prof + ((size_t)blockIdx.x * 2 + grp) * 6;
        for (int i = 0; i < 6; ++i) o[i] = tp[i + 1] - tp[i];
    }
#endif
#undef WF_TP
}

template <int K, int PRO, int SWAP, int STATS, int CHAIN>
hipError_t wf_launch(const WfParams& p, int batch, hipStream_t stream) {
    int maxm = p.side[0].M;
    if (p.nside > 1 && p.side[1].M > maxm) maxm = p.side[1].M;
    const int row_tiles = (maxm + WF_TM - 1) / WF_TM;
    constexpr size_t lds = (size_t)2 * WF_TM * (2 * K + 16) + 8 * WF_TBUF;   // half planes + transposition buffers (the norm constants borrow the latter)
    static_assert(8 * WF_TBUF >= 2 * 512 * 4, "the norm constants live in the transposition area");
    if (hipError_t e = imp_grant_dynamic_lds((const void*)gemm_wf_kernel<K, PRO, SWAP, STATS, CHAIN>, lds)) return e;
    const int psplit = p.pass_split > 1 ? p.pass_split : 1;
    if ((p.N >> 7) % psplit) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_wf_kernel<K, PRO, SWAP, STATS, CHAIN>), dim3(batch * p.nside * row_tiles * psplit), dim3(512), lds, stream, p, row_tiles);
#ifdef WF_PROFILE
    if (!CHAIN) {
        static int calls = 0;
        if (++calls % 21 == 0) {
            (void)hipStreamSynchronize(stream);
            const int nb = batch * p.nside * row_tiles * psplit < 2048 ? batch * p.nside * row_tiles * psplit : 2048;
            std::vector<unsigned long long> h((size_t)nb * 8);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(wf_prof), h.size() * 8);
            double a[2][4] = {};
            for (int i = 0; i < nb; ++i) for (int g = 0; g < 2; ++g) for (int j = 0; j < 4; ++j) a[g][j] += (double)h[((size_t)i * 2 + g) * 4 + j] / nb;
            for (int g = 0; g < 2; ++g)
                fprintf(stderr, "[gemm_wf<%d,%d,%d,%d> N=%d, %d workgroups, group %d] mean cycles (100 MHz-class counter): staging %.0f  K loops %.0f  epilogues %.0f  total %.0f\n",
                        K, PRO, SWAP, STATS, p.N, nb, g, a[g][0], a[g][1], a[g][2], a[g][3]);
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace

int gemm_wf_stats_rows() { return WF_TM; }

bool gemm_wf_supported(int K, int N) { return (K == 256 || K == 512) && N % 128 == 0; }

hipError_t launch_gemm_wf(const WfParams& p, int batch, hipStream_t stream) {
    const bool stats = p.side[0].out_stats != nullptr;
    const bool pro = p.side[0].in_stats != nullptr;
    const bool chain = p.Wf2_ != nullptr;
    if (!gemm_wf_supported(p.K, p.N)) return hipErrorInvalidValue;
    if (chain) {
        // the chained projection re-uses the first GEMM's output tile as its A operand: all 256 columns must be in this workgroup
        if (!pro || stats || p.K != 512 || p.N != 256 || p.pass_split > 1 || !gemm_wf_supported(256, p.N2) || !p.side[0].C2) return hipErrorInvalidValue;
        return wf_launch<512, 1, 1, 0, 1>(p, batch, stream);
    }
    if (stats) {
        if (pro) return hipErrorInvalidValue;
        if (p.N > 1024 || (p.side[0].fin_stats && (!p.stat_cnt || p.N > 512))) return hipErrorInvalidValue;
        return p.K == 256 ? wf_launch<256, 0, 0, 1, 0>(p, batch, stream) : wf_launch<512, 0, 0, 1, 0>(p, batch, stream);
    }
    if (pro) return p.K == 256 ? wf_launch<256, 1, 1, 0, 0>(p, batch, stream) : wf_launch<512, 1, 1, 0, 0>(p, batch, stream);
    return p.K == 256 ? wf_launch<256, 0, 1, 0, 0>(p, batch, stream) : wf_launch<512, 0, 1, 0, 0>(p, batch, stream);
}

hipError_t launch_gemm_wf_fused(const WfParams& p, const WfFused& f, int batch, hipStream_t stream) {
    if (p.K != 512 || p.N != 512 || !p.Wf_ || !f.Wf3_ || !f.tag || !f.status || p.pass_split > 1 || p.nside < 1 || p.nside > 2) return hipErrorInvalidValue;
    for (int s = 0; s < p.nside; ++s)
        if (!f.rec[s] || !f.fin[s] || !p.side[s].A || !p.side[s].C || p.side[s].M < 1) return hipErrorInvalidValue;
    if (p.Wf2_ && (!gemm_wf_supported(256, p.N2) || !p.side[0].C2)) return hipErrorInvalidValue;
    int maxm = p.side[0].M;
    if (p.nside > 1 && p.side[1].M > maxm) maxm = p.side[1].M;
    const int row_tiles = (maxm + WF_TM - 1) / WF_TM;
    constexpr size_t lds = (size_t)2 * WF_TM * (2 * 512 + 16) + 8 * WF_TBUF;
    static_assert(8 * WF_TBUF >= 2 * 512 * 4, "(mean, rstd) of the 512 hidden channels live in the transposition area");
    if (hipError_t e = imp_grant_dynamic_lds((const void*)gemm_wf_fused_kernel, lds)) return e;
    hipLaunchKernelGGL(gemm_wf_fused_kernel, dim3(batch * p.nside * row_tiles), dim3(512), lds, stream, p, f, row_tiles);
    return hipGetLastError();
}

// W [N][K] fp32 -> MFMA fragment order [N / 32][K / 16][hi | lo][64 lanes][8 halves]: lane (col = lane % 32, k-half = lane / 32)
// holds the 8 consecutive k = 16 ks + 8 (lane / 32) ... of output column 32 nt + lane % 32
void wf_pack(const float* W, int N, int K, _Float16* out) {
    const int ksteps = K / 16;
    for (int nt = 0; nt < N / 32; ++nt)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const float v = W[(size_t)(nt * 32 + (lane & 31)) * K + ks * 16 + 8 * (lane >> 5) + e];
                    const _Float16 hi = (_Float16)v;
                    const size_t base = ((size_t)(nt * ksteps + ks) * 2) * 64 * 8;
                    out[base + (size_t)lane * 8 + e] = hi;
                    out[base + 64 * 8 + (size_t)lane * 8 + e] = (_Float16)(v - (float)hi);
                }
}
