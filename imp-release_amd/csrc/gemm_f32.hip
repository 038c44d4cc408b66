// fp32-input MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains on the matrix pipe,
// 157 TFLOP/s peak = 1/16 of bf16 MFMA, so operand feed from LDS is never the limiter).
//
//   C[M x N] = epilogue( prologue(A)[M x K] * W[N x K]^T )
//
// Every 1x1 Conv1d of the reference (nets/layers.py:59-77,106-107,119,134,195-196; nets/gm.py:69-72)
// is this GEMM on token-major activations; the score matrix of nets/gm.py:293-294 and the
// re-materialised attention probabilities are the same kernel with per-batch "weights".
//
// Workgroup = 4 wave64 as 2(M) x 2(N); wave tile (BM/2) x (BN/2) built from 32x32 MFMA tiles; BK = 32.
// Operands are staged global -> registers -> (normalise + activation) -> LDS.  ONE LDS buffer (37 KB for
// 128x128) + register prefetch of the next K-tile: the loads of tile k+1 are in flight while tile k is
// multiplied, and 3 workgroups fit a CU, which is what hides the two barriers per K-tile (fp32 MFMA is
// 64 cycles per instruction, so one wave per SIMD already saturates the pipe when it never stalls; the
// co-resident workgroups cover the stalls).  LDS rows are padded to 36 floats so that the ds_read_b128
// fragment reads of 16 consecutive rows hit 16 distinct 16-byte slots (conflict-free).
//
// The K-loop is branch-free: out-of-range rows of A / W are CLAMPED to the last valid row instead of being
// predicated (a garbage A row only feeds an output row >= M and a garbage W row an output column >= N,
// neither of which is ever stored or counted), row pointers are computed once, and the prologue mode is a
// template parameter (PRO 0: none, 1: InstanceNorm/fixed norm + ReLU, 2: generic affine norm + any activation).
//
// PREC = 1 ("f16x3"): every fp32 operand x is split on the fly into two halves, hi = f16(x) and lo = f16(x - hi)
// (22 significant bits together), and each fp32 product becomes three f16 MFMAs (hi.hi + hi.lo + lo.hi, fp32
// accumulate, v_mfma_f32_32x32x16_f16): fp32-level results (validated end to end against the reference fixtures)
// at 3/16 of the fp32-MFMA pipe time.  The LDS tile keeps its size: a row holds [32 hi halves | 32 lo halves].
//
// MFMA k-pairing (fp32 path): step s of a 32-deep K-tile multiplies k = s (lanes 0-31) and k = 16 + s (lanes 32-63);
// A and W fragments use the same pairing so each lane reads 16 contiguous floats of its row.
#include "imp_kernels.h"
#include <stdlib.h>
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

#ifdef GEMM_PROFILE   // tools/probe/gemm_probe.hip: cycle stamps of the K-loop phases of workgroup 0 (f16x3 path)
__device__ unsigned long long gemm_prof[4][8];
#define GP_CLK(i) { const unsigned long long c_ = __builtin_readcyclecounter(); prof[i] += c_ - tlast; tlast = c_; }
#else
#define GP_CLK(i)
#endif

constexpr int BK = 32;
constexpr int LDT = BK + 4;   // padded LDS row (floats)

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 0) return fmaxf(v, 0.0f);                       // relu
    if (act == 2) return v > 0.0f ? v : 0.1f * v;              // leaky relu(0.1)
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); // exact gelu
}

// bijective XCD-aware remap of a linear block id (hardware places block b on XCD b % 8): each XCD gets a
// contiguous chunk of the logical tile order, in which the column tiles sharing an A row-tile are adjacent.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = lin % 8, idx = lin / 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// x -> (hi, lo) halves, round-to-nearest both times; x - float(hi) is exact in fp32 (imp_split2, imp_kernels.h)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { unsigned a, b; imp_split2(v[2 * i], v[2 * i + 1], a, b); hi[i] = a; lo[i] = b; }
}

template <int BM, int BN, int PRO, int PREC, int DEEP>
__global__ __launch_bounds__(256, DEEP ? 2 : 3) void gemm_f32_kernel(const GemmParams p, int col_tiles, int row_tiles, int total) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int LA = BM / 32, LW = BN / 32;   // float4 loads per thread per K-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [BM][LDT]
    float* Ws = smem + BM * LDT;            // [BN][LDT]
    float* tr = Ws + BN * LDT;              // [mu K][rs K]([gamma K][beta K])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int z = xcd_remap(blockIdx.x, total);
    const int ctile = z % col_tiles; z /= col_tiles;
    const int rtile = z % row_tiles; z /= row_tiles;
    const int sidx = z % p.nside; z /= p.nside;
    const int sub = z % p.nsub;
    const int b = z / p.nsub;
    const GemmSide& S = p.side[sidx];
    const int Mpad = S.M;                                   // strides and the statistics layout follow the padded size
    const int M = imp_count(p.rc, S.img, b, Mpad), N = S.N, K = p.K;      // (ragged batches: this pair's own row count; 0 = retired pair)
    const int row0 = rtile * BM, col0 = ctile * BN;
    if (row0 >= M || col0 >= N) return;     // uniform per workgroup, before any barrier

    const float* A = S.A + b * S.sA_b + sub * S.sA_s;
    const float* A2 = S.A2 ? S.A2 + b * S.sA_b + sub * S.sA_s : A;
    const float* W = S.W + b * S.sW_b + sub * S.sW_s;
    const int flags = p.flags;

    // ---- prologue: per-channel normalisation constants -------------------------------------------
    if (PRO != 0) {
        for (int k = tid; k < K; k += 256) {
            float mu, rs;
            if (S.in_stats) {
                const float* st = S.in_stats + ((long)b * K + k) * 2;
                mu = st[0];
                rs = st[1];
            } else {
                mu = p.nm_mean[k];
                rs = p.nm_rstd[k];
            }
            tr[k] = mu;
            tr[K + k] = rs;
            if (PRO == 2) {
                tr[2 * K + k] = (flags & GEMM_PRO_AFFINE) ? p.nm_gamma[k] : 1.f;
                tr[3 * K + k] = (flags & GEMM_PRO_AFFINE) ? p.nm_beta[k] : 0.f;
            }
        }
        __syncthreads();
    }

    // ---- per-thread staging geometry (loop invariant) ----------------------------------------------
    const int lr = tid >> 3, lc = (tid & 7) << 2;   // row within a 32-row group, float offset in the K-tile
    const float* pa[LA];
    const float* pa2[LA];
    const float* pw[LW];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int r = min(row0 + lr + 32 * j, M - 1);
        pa[j] = A + (long)r * p.lda + lc;
        pa2[j] = A2 + (long)r * p.lda2 + lc - p.ksplit;
    }
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int r = min(col0 + lr + 32 * j, N - 1);
        pw[j] = W + (long)r * p.ldw + lc;
    }
    // staged tile(s) in registers: one set (prefetch distance 1, 3 workgroups per CU) or, DEEP, two sets that alternate
    // (a tile is requested two K-steps before it is converted; 2 workgroups per CU - for launches that have no more)
    f32x4 ra0[LA], rw0[LW], ra1[DEEP ? LA : 1], rw1[DEEP ? LW : 1];
    auto load_tile = [&](int kt, f32x4 (&ra)[LA], f32x4 (&rw)[LW]) {
        const int k0 = kt * BK;
        const bool second = k0 >= p.ksplit;          // uniform
#pragma unroll
        for (int j = 0; j < LA; ++j) ra[j] = *reinterpret_cast<const f32x4*>((second ? pa2[j] : pa[j]) + k0);
#pragma unroll
        for (int j = 0; j < LW; ++j) rw[j] = *reinterpret_cast<const f32x4*>(pw[j] + k0);
    };
    auto store_tile = [&](int kt, f32x4 (&ra)[LA], f32x4 (&rw)[LW]) {
        if (PRO != 0) {
            const int k0 = kt * BK + lc;
            const f32x4 mu = *reinterpret_cast<const f32x4*>(tr + k0);
            const f32x4 rs = *reinterpret_cast<const f32x4*>(tr + K + k0);
            if (PRO == 1) {
#pragma unroll
                for (int j = 0; j < LA; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra[j][e] = fmaxf((ra[j][e] - mu[e]) * rs[e], 0.f);
            } else {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(tr + 2 * K + k0);
                const f32x4 be = *reinterpret_cast<const f32x4*>(tr + 3 * K + k0);
                const int act = p.act;
#pragma unroll
                for (int j = 0; j < LA; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = (ra[j][e] - mu[e]) * rs[e];
                        if (flags & GEMM_PRO_AFFINE) v = v * ga[e] + be[e];
                        ra[j][e] = apply_act(v, act);
                    }
            }
        }
        if (PREC == 0) {
#pragma unroll
            for (int j = 0; j < LA; ++j) *reinterpret_cast<f32x4*>(As + (lr + 32 * j) * LDT + lc) = ra[j];
#pragma unroll
            for (int j = 0; j < LW; ++j) *reinterpret_cast<f32x4*>(Ws + (lr + 32 * j) * LDT + lc) = rw[j];
        } else {
            // row = [hi: 32 halves = 16 dwords | lo: 32 halves | 4 dwords pad]; this thread owns k = lc..lc+3
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                u32x2 hi, lo;
                split4(ra[j], hi, lo);
                float* row = As + (lr + 32 * j) * LDT;
                *reinterpret_cast<u32x2*>(row + (lc >> 1)) = hi;
                *reinterpret_cast<u32x2*>(row + 16 + (lc >> 1)) = lo;
            }
#pragma unroll
            for (int j = 0; j < LW; ++j) {
                u32x2 hi, lo;
                split4(rw[j], hi, lo);
                float* row = Ws + (lr + 32 * j) * LDT;
                *reinterpret_cast<u32x2*>(row + (lc >> 1)) = hi;
                *reinterpret_cast<u32x2*>(row + 16 + (lc >> 1)) = lo;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = K / BK;
    load_tile(0, ra0, rw0);
    const int frow = lane & 31;
    if (PREC == 0) {
        const int fk = (lane >> 5) * 16;
        const float* as = As + (wm * WM + frow) * LDT + fk;
        const float* ws = Ws + (wn * WN + frow) * LDT + fk;
        for (int kt = 0; kt < nkt; ++kt) {
            store_tile(kt, ra0, rw0);
            __syncthreads();
            load_tile(kt + 1 < nkt ? kt + 1 : kt, ra0, rw0);   // in flight during the MFMAs below (last one: harmless re-load)
            f32x4 af[2][TM], wf[2][TN];              // fragment double buffer: c+1 is read while c is multiplied
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDT);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[0][j] = *reinterpret_cast<const f32x4*>(ws + j * 32 * LDT);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[(c + 1) & 1][i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDT + (c + 1) * 4);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        wf[(c + 1) & 1][j] = *reinterpret_cast<const f32x4*>(ws + j * 32 * LDT + (c + 1) * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][i][e], wf[c & 1][j][e], acc[i][j], 0, 0, 0);
            }
            __syncthreads();                          // every wave is done reading before the next store_tile
        }
    } else {
        // f16x3: k-step s (16 deep) of the tile; lane (row, half) reads 8 consecutive halves k = 16 s + 8 half ...
        const float* as = As + (wm * WM + frow) * LDT + (lane >> 5) * 4;
        const float* ws = Ws + (wn * WN + frow) * LDT + (lane >> 5) * 4;
        constexpr int DIST = DEEP ? 2 : 1;
#ifdef GEMM_PROFILE
        unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long tlast = __builtin_readcyclecounter();
#endif
        auto k_step = [&](int kt, f32x4 (&ra)[LA], f32x4 (&rw)[LW]) {
            GP_CLK(5);
            store_tile(kt, ra, rw);
            GP_CLK(0);
            __syncthreads();
            GP_CLK(1);
            load_tile(kt + DIST < nkt ? kt + DIST : nkt - 1, ra, rw);   // (past the end: harmless re-load)
            GP_CLK(2);
            f16x8 ah[2][TM], al[2][TM], wh[2][TN], wl[2][TN];
#pragma unroll
            for (int sK = 0; sK < 2; ++sK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[sK][i] = *reinterpret_cast<const f16x8*>(as + i * 32 * LDT + sK * 8);
                    al[sK][i] = *reinterpret_cast<const f16x8*>(as + i * 32 * LDT + 16 + sK * 8);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    wh[sK][j] = *reinterpret_cast<const f16x8*>(ws + j * 32 * LDT + sK * 8);
                    wl[sK][j] = *reinterpret_cast<const f16x8*>(ws + j * 32 * LDT + 16 + sK * 8);
                }
            }
            // product-major issue order: consecutive MFMAs write DIFFERENT accumulators (no dependent back-to-back pairs)
#pragma unroll
            for (int sK = 0; sK < 2; ++sK) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sK][i], wh[sK][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sK][i], wl[sK][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sK][i], wh[sK][j], acc[i][j], 0, 0, 0);
            }
            GP_CLK(3);
            __syncthreads();
            GP_CLK(4);
        };
        if (DEEP) {
            if constexpr (DEEP != 0) {
                load_tile(nkt > 1 ? 1 : 0, ra1, rw1);
                for (int kt = 0; kt < nkt; kt += 2) {
                    k_step(kt, ra0, rw0);
                    if (kt + 1 < nkt) k_step(kt + 1, ra1, rw1);
                }
            }
        } else {
            for (int kt = 0; kt < nkt; ++kt) k_step(kt, ra0, rw0);
        }
#ifdef GEMM_PROFILE
        if (blockIdx.x == 0 && lane == 0)
            for (int i = 0; i < 8; ++i) gemm_prof[wave][i] = prof[i];
#endif
    }

    // ---- epilogue: value transforms with the (uniform) flags hoisted out of the element loops -------------
    const int half = lane >> 5;
    const int rbase = row0 + wm * WM + 4 * half;            // + i*32 + (r&3) + 8*(r>>2)
    const int cbase = col0 + wn * WN + (lane & 31);         // + j*32
    if (flags & GEMM_EPI_DIV) {
        const float dv = p.div;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] / dv;
    }
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bv = p.bias[min(cbase + j * 32, N - 1)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
        }
    }
    if (flags & GEMM_EPI_EXPROW) {
        const float* rv = S.rowvec + b * S.sRV_b + sub * S.sRV_s;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lv = rv[min(rbase + i * 32 + (r & 3) + 8 * (r >> 2), M - 1)];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = expf(acc[i][j][r] - lv);
            }
    }
    if (S.R) {
        const float* R = S.R + b * S.sR_b;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long ro = (long)min(rbase + i * 32 + (r & 3) + 8 * (r >> 2), M - 1) * p.ldr;
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] += R[ro + min(cbase + j * 32, N - 1)];
            }
    }
    const bool interior = row0 + BM <= M && col0 + BN <= N;     // uniform
    if (!(flags & GEMM_EPI_NOSTORE)) {
        float* C = S.C + b * S.sC_b + sub * S.sC_s;
        if (interior) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* crow = C + (long)(rbase + i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc + cbase;
#pragma unroll
                    for (int j = 0; j < TN; ++j) crow[j * 32] = acc[i][j][r];
                }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (row < M && cbase + j * 32 < N) C[(long)row * p.ldc + cbase + j * 32] = acc[i][j][r];
                }
        }
    }
    if (flags & GEMM_EPI_STATS) {
        // per column and per 32-ROW BLOCK (one MFMA row tile of the wave = one statistics block): (sum, M2 about the block mean) over the valid
        // rows, two passes over the accumulators: in-lane over the 16 registers, then the lane pair (l, l ^ 32).  No
        // LDS, no cross-wave step; launch_stats_finalize merges the blocks with Chan's formula in fp64, so a channel
        // whose |mean| is far above its standard deviation keeps its digits (E[x^2] - mean^2 would cancel them).
        // Round 6: the block is 32 rows for EVERY tile shape (it was the wave's WM = 32 or 64 rows: the tile - chosen from the launch's workgroup
        // count, i.e. from the batch - decided how a pair's rows were grouped, and the InstanceNorm statistics moved in the last bits with it)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int blk_row0 = row0 + wm * WM + i * 32;
            const int cnt = min(32, M - blk_row0);                    // valid rows of this block (<= 0: nothing to report)
            if (cnt <= 0) continue;                                   // (uniform per wave)
            const int tiles_side = (Mpad + 31) / 32;                  // dense per side: [b][block][N][2]
            float* os = S.out_stats + ((long)b * tiles_side + ((rtile * 2 + wm) * TM + i)) * N * 2;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = interior || (rbase + i * 32 + (r & 3) + 8 * (r >> 2) < M);
                    s += ok ? acc[i][j][r] : 0.f;
                }
                s += __shfl_xor(s, 32);
                const float mean = s / (float)cnt;
                float m2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = interior || (rbase + i * 32 + (r & 3) + 8 * (r >> 2) < M);
                    const float d = acc[i][j][r] - mean;
                    m2 = fmaf(ok ? d : 0.f, d, m2);
                }
                m2 += __shfl_xor(m2, 32);
                const int col = cbase + j * 32;
                if (lane < 32 && col < N) {
                    os[(long)col * 2] = s;
                    os[(long)col * 2 + 1] = m2;
                }
            }
        }
    }
}

// per-block (sum, M2 about the block mean) -> (mean, rstd): Chan's parallel merge, fp64, fixed order.
// block t covers rows [t * tile_rows, min(M, (t + 1) * tile_rows)).  With n_t rows, sum s_t and M2_t per block:
//     mean = sum_t s_t / N,   M2 = sum_t M2_t + sum_t s_t^2 / n_t - N mean^2
// (the last two terms are the between-block part sum_t n_t (mean_t - mean)^2; in fp64 their cancellation costs
// (mean / std)^2 x 2^-53 relative - nothing for any fp32-representable channel).  Workgroup = 64 channels x 4 block groups
// (group g takes blocks g, g + 4, ...; 4 independent loads in flight), combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void stats_finalize_kernel(StatsSide s0, StatsSide s1, int K, float eps, RaggedCounts rc) {
    __shared__ double sm[4][3][64];
    const StatsSide& S = blockIdx.y == 0 ? s0 : s1;
    const int b = blockIdx.z;
    const int M = imp_count(rc, blockIdx.y, b, S.M);                          // ragged batches: this pair's own row count (the sides are image 0, image 1)
    if (M <= 0) return;                                                       // retired pair
    const int tiles = rc.on ? (M + S.tile_rows - 1) / S.tile_rows : S.tiles;  // blocks that were written; the layout keeps S.tiles per pair
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + cx;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (k < K) {
        const float2* st = reinterpret_cast<const float2*>(S.part) + (long)b * S.tiles * K + k;
        int t = g;
        for (; t + 12 < tiles; t += 16) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = st[(long)(t + 4 * u) * K];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nt = min(S.tile_rows, M - (t + 4 * u) * S.tile_rows);
                a1 += (double)v[u].x; a2 += (double)v[u].y; a3 += (double)v[u].x * (double)v[u].x / (double)nt;
            }
        }
        for (; t < tiles; t += 4) {
            const float2 v = st[(long)t * K];
            const int nt = min(S.tile_rows, M - t * S.tile_rows);
            a1 += (double)v.x; a2 += (double)v.y; a3 += (double)v.x * (double)v.x / (double)nt;
        }
    }
    sm[g][0][cx] = a1; sm[g][1][cx] = a2; sm[g][2][cx] = a3;
    __syncthreads();
    if (g == 0 && k < K) {
        const double s1t = (sm[0][0][cx] + sm[1][0][cx]) + (sm[2][0][cx] + sm[3][0][cx]);
        const double m2w = (sm[0][1][cx] + sm[1][1][cx]) + (sm[2][1][cx] + sm[3][1][cx]);
        const double sqn = (sm[0][2][cx] + sm[1][2][cx]) + (sm[2][2][cx] + sm[3][2][cx]);
        const double mean = s1t / (double)M;
        double m2 = m2w + (sqn - (double)M * mean * mean);
        m2 = m2 < 0.0 ? 0.0 : m2;
        float2 o;
        o.x = (float)mean;
        o.y = (float)(1.0 / sqrt(m2 / (double)M + (double)eps));       // biased variance, eps inside the root (nets/layers.py:67-68)
        reinterpret_cast<float2*>(S.out)[(long)b * K + k] = o;
    }
}

size_t gemm_lds_bytes(int BM, int BN, int pro, int K) {
    size_t f = (size_t)(BM + BN) * LDT;
    if (pro) f += (size_t)K * (pro == 2 ? 4 : 2);
    return f * sizeof(float);
}

// tile choice: the largest tile that still gives >= 2 workgroups per CU (512); all three keep 4 waves
void gemm_pick_tile(int M, int N, int total_z, int* bm, int* bn) {
    auto wgs = [&](int tm, int tn) { return (long)((M + tm - 1) / tm) * ((N + tn - 1) / tn) * total_z; };
    if (wgs(128, 128) >= 512) { *bm = 128; *bn = 128; }
    else if (wgs(128, 64) >= 512) { *bm = 128; *bn = 64; }
    else { *bm = 64; *bn = 64; }
}

template <int BM, int BN, int PRO, int PREC, int DEEP>
hipError_t gemm_launch_one(const GemmParams& p, dim3 grid, hipStream_t stream) {
    const size_t lds = gemm_lds_bytes(BM, BN, PRO, p.K);
    if (hipError_t e = imp_grant_dynamic_lds((const void*)gemm_f32_kernel<BM, BN, PRO, PREC, DEEP>, lds)) return e;
    const int total = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, PRO, PREC, DEEP>), dim3(total), dim3(256), lds, stream, p, (int)grid.x, (int)grid.y,
                       total);
    return hipGetLastError();
}

template <int BM, int BN>
hipError_t gemm_launch_pro(const GemmParams& p, int pro, dim3 grid, hipStream_t stream) {
    if (p.prec == 1) {
        // two register sets (prefetch distance 2) where they are free: the 128x64 and 64x64 tiles stay under 168 VGPRs
        // with them (still 3 waves per SIMD); measured 31.0 -> 28.7 us on the MLP3 GEMM, no gain on 128x128 tiles
        const bool deep = BM * BN <= 128 * 64;
        if (deep) {
            if (pro == 0) return gemm_launch_one<BM, BN, 0, 1, 1>(p, grid, stream);
            if (pro == 1) return gemm_launch_one<BM, BN, 1, 1, 1>(p, grid, stream);
            return gemm_launch_one<BM, BN, 2, 1, 1>(p, grid, stream);
        }
        if (pro == 0) return gemm_launch_one<BM, BN, 0, 1, 0>(p, grid, stream);
        if (pro == 1) return gemm_launch_one<BM, BN, 1, 1, 0>(p, grid, stream);
        return gemm_launch_one<BM, BN, 2, 1, 0>(p, grid, stream);
    }
    if (pro == 0) return gemm_launch_one<BM, BN, 0, 0, 0>(p, grid, stream);
    if (pro == 1) return gemm_launch_one<BM, BN, 1, 0, 0>(p, grid, stream);
    return gemm_launch_one<BM, BN, 2, 0, 0>(p, grid, stream);
}

}  // namespace

int gemm_tile_m(int M, int N, int total_z) {
    int bm, bn;
    gemm_pick_tile(M, N, total_z, &bm, &bn);
    return bm;
}

// rows per statistics block (= rows of one wave's accumulator tile) of the launch the parameters will get
int gemm_stats_rows(int, int, int) { return 32; }      // (round 6: one MFMA row tile, whatever tile shape the launch gets)

hipError_t launch_gemm_f32(const GemmParams& p, int batch, hipStream_t stream) {
    const int maxM = p.nside == 2 ? (p.side[0].M > p.side[1].M ? p.side[0].M : p.side[1].M) : p.side[0].M;
    const int maxN = p.nside == 2 ? (p.side[0].N > p.side[1].N ? p.side[0].N : p.side[1].N) : p.side[0].N;
    const int total_z = batch * p.nsub * p.nside;
    if (maxM <= 0 || maxN <= 0 || total_z <= 0) return hipSuccess;
    // the tile choice must be reproducible by gemm_tile_m(): it is decided on the LARGER side's M
    int bm, bn;
    gemm_pick_tile(maxM, maxN, total_z, &bm, &bn);
    dim3 grid((maxN + bn - 1) / bn, (maxM + bm - 1) / bm, total_z);
    int pro = 0;
    if (p.flags & GEMM_PRO_NORM) pro = (!(p.flags & GEMM_PRO_AFFINE) && p.act == 0) ? 1 : 2;
    if (bm == 128 && bn == 128) return gemm_launch_pro<128, 128>(p, pro, grid, stream);
    if (bm == 128 && bn == 64) return gemm_launch_pro<128, 64>(p, pro, grid, stream);
    return gemm_launch_pro<64, 64>(p, pro, grid, stream);
}

hipError_t launch_stats_finalize(const StatsSide sides[2], int nside, int batch, int K, float eps, hipStream_t stream, const RaggedCounts* rc) {
    RaggedCounts r;
    if (rc) r = *rc; else r.on = 0;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((K + 63) / 64, nside, batch), dim3(256), 0, stream, sides[0],
                       sides[nside - 1], K, eps, r);
    return hipGetLastError();
}
