// Split-half ("f16x3") MFMA GEMM on PRE-SPLIT operands, gfx950.
//
//   C[M x N] = epilogue( A[M x K] * W[N x K]^T )          (every 1x1 Conv1d of a GNN layer: nets/layers.py:119,134,145-149,210-218)
//
// gemm_f32.hip splits every fp32 operand into (hi, lo) halves while staging it - every workgroup re-converts the same
// weights and the same activations N / BN times, and the conversion (3 VALU per pair of values + ds_write) sits between
// the global load and the MFMAs.  Here the operands live in memory as "planes": a row of a C-channel matrix is
// [C hi halves | C lo halves] (hi = f16(x), lo = f16(x - hi); same 4 bytes per element as fp32), written once by the
// producer (weights: at pack time; activations: by the producing kernel's epilogue).  Staging is then a pure copy, done by
// the LDS-DMA path (global_load_lds_dwordx4: no staging registers, no ds_write pass), double buffered, ONE workgroup
// barrier per 32-deep K-tile, and the K-loop contains nothing but ds_read_b128 + MFMA:
//     per k16-step and wave: 8 ds_read_b128 (2 A tiles + 2 W tiles, hi and lo) feed 12 MFMAs (lo.hi + hi.lo + hi.hi).
//
// LDS image of one operand tile (R rows x 32 k, two planes): pieces of 16 rows x 64 B = 1 KiB, written lane-linearly by
// one DMA instruction.  Lane l of the DMA fetches row l / 4, 16-byte k-chunk (l % 4) ^ ((row / 4) % 4) - four lanes
// read one contiguous 64-byte row segment - so the chunk c of row i sits in slot 4 i + (c ^ (i / 4 % 4)): the ds_read_b128
// fragment reads (16 rows x one chunk per lane group) hit 16 distinct 16-byte slots of the 256-byte bank row
// (conflict-free), and the two k16-steps of a tile differ by XOR 32 in the byte address.
//
// ASRC = 1 (the MLP's second conv): A is the fp32 hidden activation; InstanceNorm / BatchNorm + activation are applied
// while it is staged through registers (next K-tile prefetched) and split into the same LDS image; W still comes by DMA.
//
// Epilogue through LDS (the two stages are free by then): + bias, per-column-range scale (the softmax scale of the query
// section), exp(. - rowvec), + fp32 residual, per-tile column statistics (sum, M2 about the tile mean: merged with
// Chan's formula, so a channel whose |mean| >> std loses no digits), fp32 output rows as float4, and / or the result
// as planes for the next consumer.
#include "../../imp-release_amd/csrc/imp_kernels.h"
// RETIRED (round 3): measured slower than gemm_f32.hip / gemm_wf.hip on MI355X ("a negative result"); no longer
// part of libimp_hip.so.  Its interface, formerly in imp_kernels.h, lives here so that the file still compiles on its own:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/probe/gemm_planes.hip
enum {
    PG_PRO_AFFINE = 1 << 1,       // fp32-A prologue: norm has gamma/beta (BatchNorm)
    PG_EPI_EXPROW = 1 << 2,       // v = exp(v - rowvec[row] * rowvec_scale)
    PG_EPI_STATS = 1 << 3,        // per-tile per-column (sum, M2 about the tile mean) -> out_stats [b][row_tiles][N][2]
    PG_EPI_EXP2 = 1 << 4,         // EXPROW in base 2
};
struct PGemmSide {
    const _Float16* Ap;    // A planes [b][sub][M][lda halves], lo plane at + apw halves; null when Af is given
    const _Float16* Ap2;   // optional second K-range source (k >= ksplit): [b][M][lda2], lo at + apw2
    const float* Af;       // A as fp32 [b][M][ldaf]: normalised + activated + split while staging (in_stats or nm_*)
    const _Float16* Wp;    // W planes [b][sub][N][ldw halves], lo plane at + K
    float* C;              // fp32 output [b][sub][M][ldc] or null
    _Float16* Cp;          // planes output [b][M][ldcp halves] in groups of cpw channels ([hi cpw | lo cpw] per group) or null
    const float* R;        // fp32 residual [b][M][ldr] or null
    const float* rowvec;   // [b][sub][M] or null
    const float* in_stats; // [b][K][2] finalised (mean, rstd) or null
    float* out_stats;      // [b][row_tiles][N][2]
    long sA_b, sA_s, sA2_b, sAf_b, sW_b, sW_s, sC_b, sC_s, sCp_b, sR_b, sRV_b, sRV_s;
    int M, N;
};
struct PGemmParams {
    PGemmSide side[2];
    const float* bias;
    const float *nm_mean, *nm_rstd, *nm_gamma, *nm_beta;
    int K, ksplit;
    int lda, lda2, apw, apw2, ldaf, ldw, ldc, ldcp, cpw, ldr;
    int nside, nsub, flags, act;
    int bn_hint;           // 64 forces 64-column tiles
    int scale_cols;        // output columns < scale_cols are multiplied by `scale` after the bias (multiple of 128)
    float scale, rowvec_scale;
    int dbg;               // probe switches (gemm_planes.hip), 0 in the product
};
hipError_t launch_gemm_planes(const PGemmParams& p, int batch, hipStream_t stream);
int pgemm_stats_rows(const PGemmParams& p, int batch);   // rows per statistics tile of the kernel that will run
hipError_t launch_make_planes(const float* x, _Float16* out, long rows, int C, long ldx, long ldo, hipStream_t stream);

#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 128, BK = 32;

__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = lin % 8, idx = lin / 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 0) return fmaxf(v, 0.0f);
    if (act == 2) return v > 0.0f ? v : 0.1f * v;
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
}
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// slot (in 16-byte units) of k-chunk c of row i inside a 16-row piece
__device__ __forceinline__ int slot_of(int i, int c) { return 4 * i + (c ^ ((i >> 2) & 3)); }

template <int BN, int ASRC, int PRO>     // PRO (ASRC = 1 only): 1 = InstanceNorm/fixed norm + ReLU, 2 = generic affine norm + any activation
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(const PGemmParams p, int col_tiles, int row_tiles, int total) {
    constexpr int TN = BN / 64;                       // 32-column MFMA tiles per wave (wave tile 64 x BN/2)
    constexpr int A_BYTES = BM * 64 * 2;              // one stage of A: BM rows x 64 B x 2 planes
    constexpr int W_BYTES = BN * 64 * 2;
    constexpr int STAGE = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* tr = reinterpret_cast<float*>(smem + 2 * STAGE);        // ASRC = 1: [mu K][rs K]([gamma K][beta K])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int z = xcd_remap(blockIdx.x, total);
    const int ctile = z % col_tiles; z /= col_tiles;
    const int rtile = z % row_tiles; z /= row_tiles;
    const int sidx = z % p.nside; z /= p.nside;
    const int sub = z % p.nsub;
    const int b = z / p.nsub;
    const PGemmSide& S = p.side[sidx];
    const int M = S.M, N = S.N, K = p.K;
    const int row0 = rtile * BM, col0 = ctile * BN;
    if (row0 >= M || col0 >= N) return;               // uniform, before any barrier

    // ---- operand bases ------------------------------------------------------------------------------------------------
    // planes rows: [K hi | K lo] halves per source (A, A2: k >= ksplit, its own row pitch); W rows: [K hi | K lo]
    const _Float16* Ap = S.Ap + b * S.sA_b + sub * S.sA_s;
    const _Float16* Ap2 = S.Ap2 ? S.Ap2 + b * S.sA2_b : Ap;
    const float* Af = S.Af ? S.Af + b * S.sAf_b : nullptr;
    const _Float16* Wp = S.Wp + b * S.sW_b + sub * S.sW_s;
    const int flags = p.flags;
    const int dbg = p.dbg;                          // probe switches (tools/probe): 1 no stores, 2 no MFMA, 4 no DMA after tile 0, 8 no epilogue

    if (ASRC == 1) {
        for (int k = tid; k < K; k += 256) {
            float mu, rs;
            if (S.in_stats) {
                const float* st = S.in_stats + ((long)b * K + k) * 2;
                mu = st[0]; rs = st[1];
            } else { mu = p.nm_mean[k]; rs = p.nm_rstd[k]; }
            tr[k] = mu; tr[K + k] = rs;
            if (PRO == 2) {
                tr[2 * K + k] = (flags & PG_PRO_AFFINE) ? p.nm_gamma[k] : 1.f;
                tr[3 * K + k] = (flags & PG_PRO_AFFINE) ? p.nm_beta[k] : 0.f;
            }
        }
    }

    // ---- DMA geometry: this wave issues pieces wave, wave + 4, ... of each operand ----------------------------------------
    // piece q of A: plane q & 1, 16-row block q >> 1 (BM / 16 blocks); lane: row l / 4, chunk (l % 4) ^ (row / 4 % 4)
    const int li = lane >> 2, lc = (lane & 3) ^ ((li >> 2) & 3);
    constexpr int APW = (ASRC == 0) ? (BM / 16) * 2 / 4 : 0;      // A pieces per wave and stage
    constexpr int WPW = (BN / 16) * 2 / 4;
    const char* asrc[APW > 0 ? APW : 1];
    const char* asrc2[APW > 0 ? APW : 1];
    const char* wsrc[WPW];
    if (ASRC == 0) {
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int q = wave + 4 * j, plane = q & 1, blk = q >> 1;
            const int r = min(row0 + blk * 16 + li, M - 1);
            asrc[j] = reinterpret_cast<const char*>(Ap + (long)r * p.lda + plane * p.apw) + lc * 16;
            asrc2[j] = reinterpret_cast<const char*>(Ap2 + (long)r * p.lda2 + plane * p.apw2) + lc * 16 - (long)p.ksplit * 2;
        }
    }
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int q = wave + 4 * j, plane = q & 1, blk = q >> 1;
        const int r = min(col0 + blk * 16 + li, N - 1);
        wsrc[j] = reinterpret_cast<const char*>(Wp + (long)r * p.ldw + plane * K) + lc * 16;
    }
    auto dma_stage = [&](int kt, int stage) {
        char* sbase = smem + stage * STAGE;
        const long koff = (long)kt * (BK * 2);
        if (ASRC == 0) {
            const bool second = kt * BK >= p.ksplit;       // uniform
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int q = wave + 4 * j;
                __builtin_amdgcn_global_load_lds((gbl_void*)((second ? asrc2[j] : asrc[j]) + koff),
                                                 (lds_void*)(sbase + (q & 1) * (BM * 64) + (q >> 1) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int q = wave + 4 * j;
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[j] + koff),
                                             (lds_void*)(sbase + A_BYTES + (q & 1) * (BN * 64) + (q >> 1) * 1024), 16, 0, 0);
        }
    };

    // ---- ASRC = 1: fp32 A through registers (norm + activation + split), same LDS image -----------------------------------
    constexpr int LA = BM / 32;
    const int sr = tid >> 3, sk = (tid & 7) << 2;        // row within a 32-row group, k offset (4 consecutive k) in the tile
    const float* pa[LA];
    f32x4 ra[LA];
    if (ASRC == 1) {
#pragma unroll
        for (int j = 0; j < LA; ++j) pa[j] = Af + (long)min(row0 + sr + 32 * j, M - 1) * p.ldaf + sk;
    }
    auto load_a = [&](int kt) {
#pragma unroll
        for (int j = 0; j < LA; ++j) ra[j] = *reinterpret_cast<const f32x4*>(pa[j] + kt * BK);
    };
    auto store_a = [&](int kt, int stage) {
        const int k0 = kt * BK + sk;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(tr + k0);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(tr + K + k0);
        f32x4 ga, be;
        if (PRO == 2) { ga = *reinterpret_cast<const f32x4*>(tr + 2 * K + k0); be = *reinterpret_cast<const f32x4*>(tr + 3 * K + k0); }
        char* sbase = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            f32x4 v = ra[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[e] - mu[e]) * rs[e];
                if (PRO == 1) t = fmaxf(t, 0.f);
                else { if (flags & PG_PRO_AFFINE) t = t * ga[e] + be[e]; t = apply_act(t, p.act); }
                v[e] = t;
            }
            u32x2 hi, lo;
            { unsigned a, c; imp_split2(v[0], v[1], a, c); hi[0] = a; lo[0] = c; imp_split2(v[2], v[3], a, c); hi[1] = a; lo[1] = c; }
            const int row = sr + 32 * j, i = row & 15, blk = row >> 4;
            const int byte = blk * 1024 + slot_of(i, sk >> 3) * 16 + ((sk >> 2) & 1) * 8;
            *reinterpret_cast<u32x2*>(sbase + byte) = hi;
            *reinterpret_cast<u32x2*>(sbase + BM * 64 + byte) = lo;
        }
    };

    // ---- fragment addresses (bytes inside a stage) --------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    int aoff[2], woff[TN];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + fr;
        aoff[i] = (row >> 4) * 1024 + slot_of(row & 15, fh) * 16;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * (BN / 2) + j * 32 + fr;
        woff[j] = A_BYTES + (col >> 4) * 1024 + slot_of(col & 15, fh) * 16;
    }

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = K / BK;
    if (ASRC == 1) { load_a(0); __syncthreads(); store_a(0, 0); if (nkt > 1) load_a(1); }
    dma_stage(0, 0);
    __syncthreads();                                   // (drains the DMA: vmcnt(0) inside the barrier's release)
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nkt && !(dbg & 4)) dma_stage(kt + 1, st ^ 1);   // the other stage was released by the barrier that ended step kt - 1
        const char* sb = smem + st * STAGE;
        if (!(dbg & 2))
#pragma unroll
        for (int s = 0; s < 2; ++s) {                  // two k16-steps: byte address ^ 32
            f16x8 ah[2], al[2], wh[TN], wl[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(sb + (aoff[i] ^ (s * 32)));
                al[i] = *reinterpret_cast<const f16x8*>(sb + BM * 64 + (aoff[i] ^ (s * 32)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(sb + (woff[j] ^ (s * 32)));
                wl[j] = *reinterpret_cast<const f16x8*>(sb + BN * 64 + (woff[j] ^ (s * 32)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
        }
        if (ASRC == 1 && kt + 1 < nkt) {               // convert the prefetched A tile into the other stage, fetch the one after
            store_a(kt + 1, st ^ 1);
            if (kt + 2 < nkt) load_a(kt + 2);
        }
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    // value transforms in registers (lane = one column, 16 rows per accumulator), then through an XOR-swizzled fp32 LDS tile
    // T[BM][BN] (float4 index ^ (row & 7)) so that rows leave as 16-byte segments whatever the output format
    if (dbg & 8) { if (acc[0][0][0] == 123.456f) S.C[0] = 1.f; return; }
    const int half = lane >> 5;
    const int rloc = wm * 64 + 4 * half;                 // + i*32 + (r&3) + 8*(r>>2)
    const int cloc = wn * (BN / 2) + (lane & 31);        // + j*32
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bv = p.bias[min(col0 + cloc + j * 32, N - 1)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
        }
    }
    if (p.scale_cols > 0) {                               // e.g. the query section carries the softmax scale (uniform per tile)
        if (col0 < p.scale_cols) {
            const float sc = p.scale;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
        }
    }
    if (flags & PG_EPI_EXPROW) {
        const float* rv = S.rowvec + b * S.sRV_b + sub * S.sRV_s;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lv = rv[min(row0 + rloc + i * 32 + (r & 3) + 8 * (r >> 2), M - 1)] * p.rowvec_scale;
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = (flags & PG_EPI_EXP2) ? __builtin_amdgcn_exp2f(acc[i][j][r] - lv) : expf(acc[i][j][r] - lv);
            }
    }
    float* T = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rloc + i * 32 + (r & 3) + 8 * (r >> 2), col = cloc + j * 32;
                T[row * BN + (((col >> 2) ^ (row & 7)) << 2) + (col & 3)] = acc[i][j][r];
            }
    __syncthreads();
    const int rows_here = min(BM, M - row0);
    // rows out: thread handles float4 column chunk cq of rows rr, rr + 256 / (BN / 4), ...
    constexpr int CQ = BN / 4;
    const int cq = tid % CQ;
    const int col = col0 + 4 * cq;
    const float* R = S.R ? S.R + b * S.sR_b : nullptr;
    float* C = S.C ? S.C + b * S.sC_b + sub * S.sC_s : nullptr;
    _Float16* Cp = S.Cp ? S.Cp + b * S.sCp_b : nullptr;
    // planes address of output column `col`: plane groups of p.cpw channels, [hi cpw | lo cpw] each
    const int pgrp = p.cpw > 0 ? col / p.cpw : 0, pcol = p.cpw > 0 ? col - pgrp * p.cpw : 0;
    for (int rr = tid / CQ; rr < rows_here; rr += 256 / CQ) {
        f32x4 v = *reinterpret_cast<const f32x4*>(T + rr * BN + ((cq ^ (rr & 7)) << 2));
        const long grow = row0 + rr;
        if (R) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(R + grow * p.ldr + col);
            v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
        }
        if (col < N && !((dbg & 1) && v[0] != 123.456f)) {
            if (C) *reinterpret_cast<f32x4*>(C + grow * p.ldc + col) = v;
            if (Cp) {
                u32x2 hi, lo;
                { unsigned a, c; imp_split2(v[0], v[1], a, c); hi[0] = a; lo[0] = c; imp_split2(v[2], v[3], a, c); hi[1] = a; lo[1] = c; }
                _Float16* dst = Cp + grow * p.ldcp + pgrp * 2 * p.cpw + pcol;
                *reinterpret_cast<u32x2*>(dst) = hi;
                *reinterpret_cast<u32x2*>(dst + p.cpw) = lo;
            }
        }
    }
    if (flags & PG_EPI_STATS) {
        // per column: (sum, M2 about the tile mean) over the valid rows; two threads per column (row halves), merged
        // with Chan's formula (exact counts), written per tile; launch_stats_finalize merges the tiles the same way in fp64
        const int c = tid % BN, hsel = tid / BN;                 // BN = 128: 2 halves; BN = 64: 4 quarters
        constexpr int NH = 256 / BN;
        const int r_begin = hsel * (BM / NH), r_end = min(rows_here, (hsel + 1) * (BM / NH));
        float s = 0.f;
        for (int rr = r_begin; rr < r_end; ++rr) s += T[rr * BN + (((c >> 2) ^ (rr & 7)) << 2) + (c & 3)];
        const int cnt = max(r_end - r_begin, 0);
        const float mean = cnt > 0 ? s / (float)cnt : 0.f;
        float m2 = 0.f;
        for (int rr = r_begin; rr < r_end; ++rr) {
            const float d = T[rr * BN + (((c >> 2) ^ (rr & 7)) << 2) + (c & 3)] - mean;
            m2 = fmaf(d, d, m2);
        }
        __syncthreads();                                           // everyone is done reading T
        float* sc = reinterpret_cast<float*>(smem);                // [NH][BN][2]
        sc[(hsel * BN + c) * 2] = s;
        sc[(hsel * BN + c) * 2 + 1] = m2;
        __syncthreads();
        if (hsel == 0 && col0 + c < N) {
            float ts = 0.f, tm2 = 0.f;
            int tn = 0;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int hn = max(min(rows_here, (h + 1) * (BM / NH)) - h * (BM / NH), 0);
                if (hn == 0) continue;
                const float hs = sc[(h * BN + c) * 2], hm2 = sc[(h * BN + c) * 2 + 1];
                if (tn == 0) { ts = hs; tm2 = hm2; tn = hn; }
                else {
                    const float delta = hs / (float)hn - ts / (float)tn;
                    tm2 = tm2 + hm2 + delta * delta * ((float)tn * (float)hn / (float)(tn + hn));
                    ts += hs; tn += hn;
                }
            }
            const int tiles_side = (M + BM - 1) / BM;
            float* os = S.out_stats + (((long)b * tiles_side + rtile) * N + col0 + c) * 2;
            os[0] = ts;
            os[1] = tm2;
        }
    }
}

// ---- fp32 -> planes (rows [C hi | C lo] halves); 8 channels per thread ----------------------------------------------------
__global__ __launch_bounds__(256) void make_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ out, long rows,
                                                          int C, long ldx, long ldo) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = C / 8;
    if (t >= rows * cpr) return;
    const long r = t / cpr;
    const int c = (int)(t - r * cpr) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4);
    u32x4 hi, lo;
    { unsigned h, l; imp_split2(a[0], a[1], h, l); hi[0] = h; lo[0] = l; imp_split2(a[2], a[3], h, l); hi[1] = h; lo[1] = l;
      imp_split2(d[0], d[1], h, l); hi[2] = h; lo[2] = l; imp_split2(d[2], d[3], h, l); hi[3] = h; lo[3] = l; }
    *reinterpret_cast<u32x4*>(out + r * ldo + c) = hi;
    *reinterpret_cast<u32x4*>(out + r * ldo + C + c) = lo;
}

template <int BN, int ASRC, int PRO>
hipError_t launch_one(const PGemmParams& p, dim3 grid, hipStream_t stream) {
    size_t lds = (size_t)2 * (BM * 64 * 2 + BN * 64 * 2);
    if (ASRC == 1) lds += (size_t)p.K * (PRO == 2 ? 4 : 2) * sizeof(float);
    if (hipError_t e = imp_grant_dynamic_lds((const void*)gemm_planes_kernel<BN, ASRC, PRO>, lds)) return e;
    const int total = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL((gemm_planes_kernel<BN, ASRC, PRO>), dim3(total), dim3(256), lds, stream, p, (int)grid.x, (int)grid.y, total);
    return hipGetLastError();
}

}  // namespace

int pgemm_panel_groups(const PGemmParams& p, int batch, int* ng);
hipError_t launch_gemm_panel(const PGemmParams& p, int batch, int ng, hipStream_t stream);
int pgemm_panel_rows();
// rows per statistics tile of the kernel launch_gemm_planes will pick for these parameters
int pgemm_stats_rows(const PGemmParams& p, int batch) {
    int ng;
    return pgemm_panel_groups(p, batch, &ng) ? pgemm_panel_rows() : BM;
}

// BN: 128 when that still fills the chip twice over or N needs it, else 64 (N must be a multiple of 64)
hipError_t launch_gemm_planes(const PGemmParams& p, int batch, hipStream_t stream) {
    const int maxM = p.nside == 2 ? (p.side[0].M > p.side[1].M ? p.side[0].M : p.side[1].M) : p.side[0].M;
    const int N = p.side[0].N;
    const int total_z = batch * p.nsub * p.nside;
    if (maxM <= 0 || N <= 0 || total_z <= 0) return hipSuccess;
    if (p.K % BK || N % 4) return hipErrorInvalidValue;
    int ng;
    if (pgemm_panel_groups(p, batch, &ng)) return launch_gemm_panel(p, batch, ng, stream);
    const long wg128 = (long)((maxM + BM - 1) / BM) * ((N + 127) / 128) * total_z;
    const bool bn128 = wg128 >= 512 && p.bn_hint != 64;
    const int bn = bn128 ? 128 : 64;
    dim3 grid((N + bn - 1) / bn, (maxM + BM - 1) / BM, total_z);
    const bool a_f32 = p.side[0].Af != nullptr;
    int pro = 0;
    if (a_f32) pro = (!(p.flags & PG_PRO_AFFINE) && p.act == 0) ? 1 : 2;
    if (bn128) {
        if (!a_f32) return launch_one<128, 0, 0>(p, grid, stream);
        return pro == 1 ? launch_one<128, 1, 1>(p, grid, stream) : launch_one<128, 1, 2>(p, grid, stream);
    }
    if (!a_f32) return launch_one<64, 0, 0>(p, grid, stream);
    return pro == 1 ? launch_one<64, 1, 1>(p, grid, stream) : launch_one<64, 1, 2>(p, grid, stream);
}

hipError_t launch_make_planes(const float* x, _Float16* out, long rows, int C, long ldx, long ldo, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (C % 8) return hipErrorInvalidValue;
    const long n = rows * (C / 8);
    hipLaunchKernelGGL(make_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, out, rows, C, ldx, ldo);
    return hipGetLastError();
}

// =======================================================================================================================
// Panel kernel: the same GEMM for the shapes of a GNN layer at batch sizes that fill the chip (M >= ~6000 rows).
//
// What the probe switches of the tile kernel above showed (tools/probe/gemm_time.py, B = 4, N = 2048, QKV: 31.9 us):
// the DMA + barrier loop ALONE costs 17.4 us - eight dependent global->LDS round trips per tile with a prefetch distance
// of one K-tile, every workgroup in the same phase - MFMAs add 6 us, the epilogue 8.5 us.  So the structure is turned
// around: a workgroup owns a PANEL of 64 rows and keeps its whole A operand (64 rows x 256 k as planes = 64 KB) in LDS,
// loaded by ONE burst of DMA instructions (one latency); then it walks over all N columns.  Each of the 8 waves (two per
// SIMD: one computes while the other waits for LDS / DMA) owns all 64 rows x ITS 32 columns of every 256-column tile, so a
// wave's W operand is private: it runs its own 4-stage DMA ring (2 KB per 16-deep k-step, counted s_waitcnt vmcnt, three
// steps in flight) and the main loop contains NO workgroup barrier.  A is read from HBM exactly once, W streams from L2,
// M / 64 workgroups = one per CU at B = 4, N = 2048: one round, no tail.
//   K > 256 (the MLP convs, K = 512): two K-chunks; the accumulators of ALL N columns stay in registers (NG tiles x 2 MFMA
//   tiles x 16) while the A chunk in LDS is replaced (N <= 512); N > NG * 256 (q|k|v, N = 768) needs K <= 256 and loops over
//   groups of NG column tiles, the epilogue of a group overlapping the other waves' MFMAs.
// Epilogue straight from the accumulators (lane = column, 128-byte row segments): bias, column-range scale, fp32 residual,
// fp32 rows and / or planes, and the per-panel column statistics (sum, M2 about the panel mean; the whole column of a
// panel lives in one wave: two passes over registers, no LDS).
// =======================================================================================================================
namespace {

constexpr int PM = 64;            // panel rows
constexpr int KC_MAX = 256;       // K-chunk held in LDS
constexpr int WST = 4;            // W ring stages per wave
constexpr int PW = 8;             // waves per workgroup
constexpr int WSTAGE_BYTES = 2048;            // 32 columns x 16 k x 2 planes x 2 B

template <int NG, int ASRC, int PRO>
__global__ __launch_bounds__(512, 2) void gemm_panel_kernel(const PGemmParams p, int panels_max) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int K = p.K, KC = K < KC_MAX ? K : KC_MAX, nkt = KC / BK, nchunks = K / KC;
    char* Abuf = smem;                                   // [nkt][2 planes][4 pieces][1 KiB]
    char* Wring = smem + nkt * 8192;                     // [8 waves][WST][2 KiB]
    float* tr = reinterpret_cast<float*>(Wring + PW * WST * WSTAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int z = blockIdx.x;
    const int panel = z % panels_max; z /= panels_max;
    const int sidx = z % p.nside;
    const int b = z / p.nside;
    const PGemmSide& S = p.side[sidx];
    const int M = S.M, N = S.N;
    const int row0 = panel * PM;
    if (row0 >= M) return;
    const int flags = p.flags;

    const _Float16* Ap = S.Ap ? S.Ap + b * S.sA_b : nullptr;
    const _Float16* Ap2 = S.Ap2 ? S.Ap2 + b * S.sA2_b : Ap;
    const float* Af = S.Af ? S.Af + b * S.sAf_b : nullptr;
    const char* Wp = reinterpret_cast<const char*>(S.Wp);

    if (ASRC == 1) {
        for (int k = tid; k < K; k += 512) {
            float mu, rs;
            if (S.in_stats) { const float* st = S.in_stats + ((long)b * K + k) * 2; mu = st[0]; rs = st[1]; }
            else { mu = p.nm_mean[k]; rs = p.nm_rstd[k]; }
            tr[k] = mu; tr[K + k] = rs;
            if (PRO == 2) {
                tr[2 * K + k] = (flags & PG_PRO_AFFINE) ? p.nm_gamma[k] : 1.f;
                tr[3 * K + k] = (flags & PG_PRO_AFFINE) ? p.nm_beta[k] : 0.f;
            }
        }
    }

    // ---- W ring: this wave's 32 columns of column tile jt, one 16-deep k-step per stage --------------------------------------
    // stage image [plane][chunk (8 k)][column] of 16-byte slots: DMA lane l fetches column l & 31, chunk l >> 5; the fragment
    // read of lane (column, half) is slot half * 32 + column: 16 consecutive slots per lane group, conflict-free
    const unsigned wlane = (unsigned)(((lane & 31) * (long)p.ldw + 8 * (lane >> 5)) * 2);
    char* wring = Wring + wave * (WST * WSTAGE_BYTES);
    const int ngroups = N / (NG * 256);
    const int nks = KC / 16;                                // k-steps per chunk
    int ic = 0, ig = 0, ij = 0, ik = 0, ipos = 0;           // issue cursor: runs 3 steps ahead, wraps around harmlessly at the end
    auto issue_w = [&]() {
        const long col0 = (long)((ig * NG + ij) * 256 + wave * 32);
        const char* src = Wp + (col0 * p.ldw + ic * KC + ik * 16) * 2 + wlane;
        char* dst = wring + (ipos & (WST - 1)) * WSTAGE_BYTES;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + (long)K * 2), (lds_void*)(dst + 1024), 16, 0, 0);
        ++ipos;
        if (++ik == nks) { ik = 0; if (++ij == NG) { ij = 0; if (++ig == ngroups) { ig = 0; if (++ic == nchunks) ic = 0; } } }
    };

    // ---- A chunk -> LDS ---------------------------------------------------------------------------------------------------
    const int li = lane >> 2, lc = (lane & 3) ^ ((li >> 2) & 3);
    auto load_a_chunk = [&](int c) {
        if (ASRC == 0) {
            const bool second = c * KC >= p.ksplit;
            const _Float16* src = second ? Ap2 : Ap;
            const int ld = second ? p.lda2 : p.lda, pw = second ? p.apw2 : p.apw;
            const int kbase = second ? c * KC - p.ksplit : c * KC;
            for (int q = wave; q < nkt * 8; q += PW) {     // pieces: kt x plane x 16-row block
                const int kt = q >> 3, plane = (q >> 2) & 1, blk = q & 3;
                const int r = min(row0 + blk * 16 + li, M - 1);
                const char* g = reinterpret_cast<const char*>(src + (long)r * ld + plane * pw + kbase + kt * BK) + lc * 16;
                __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(Abuf + kt * 8192 + plane * 4096 + blk * 1024), 16, 0, 0);
            }
        } else {
            // fp32 rows: normalise + activate + split; thread = row tid / 8, 4 consecutive k, 32-k stride
            const int r = tid >> 3, kq = (tid & 7) * 4;
            const float* src = Af + (long)min(row0 + r, M - 1) * p.ldaf + c * KC;
            const int i = r & 15, blk = r >> 4;
            for (int k0 = 0; k0 < KC; k0 += 128) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (k0 + 32 * u < KC) ? *reinterpret_cast<const f32x4*>(src + k0 + 32 * u + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 32 * u + kq, kg = c * KC + k;
                    if (k >= KC) break;
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(tr + kg);
                    const f32x4 rs = *reinterpret_cast<const f32x4*>(tr + K + kg);
                    f32x4 x = v[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = (x[e] - mu[e]) * rs[e];
                        if (PRO == 1) t = fmaxf(t, 0.f);
                        else {
                            if (flags & PG_PRO_AFFINE) t = t * tr[2 * K + kg + e] + tr[3 * K + kg + e];
                            t = apply_act(t, p.act);
                        }
                        x[e] = t;
                    }
                    u32x2 hi, lo;
                    { unsigned a, d; imp_split2(x[0], x[1], a, d); hi[0] = a; lo[0] = d; imp_split2(x[2], x[3], a, d); hi[1] = a; lo[1] = d; }
                    const int kt = k >> 5, kk = k & 31;
                    char* dst = Abuf + kt * 8192 + blk * 1024 + slot_of(i, kk >> 3) * 16 + ((kk >> 2) & 1) * 8;
                    *reinterpret_cast<u32x2*>(dst) = hi;
                    *reinterpret_cast<u32x2*>(dst + 4096) = lo;
                }
            }
        }
    };

    // ---- fragment addresses -------------------------------------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = (i * 2 + (fr >> 4)) * 1024 + slot_of(fr & 15, fh) * 16;
    const int woff = (fh * 32 + fr) * 16;

    f32x16 acc[NG][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NG; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    };

    // ---- epilogue of the NG column tiles of group g ---------------------------------------------------------------------------
    const int half = lane >> 5, l31 = lane & 31;
    const int rows_here = min(PM, M - row0);
    auto epilogue = [&](int g) {
        const float* R = S.R ? S.R + b * S.sR_b : nullptr;
        float* C = S.C ? S.C + b * S.sC_b : nullptr;
        _Float16* Cp = S.Cp ? S.Cp + b * S.sCp_b : nullptr;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int col = (g * NG + j) * 256 + wave * 32 + l31;
            const float bv = p.bias ? p.bias[col] : 0.f;
            const float sc = (p.scale_cols > 0 && col < p.scale_cols) ? p.scale : 1.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = (acc[j][i][r] + bv) * sc;
            if (flags & PG_EPI_STATS) {
                // whole column of the panel in this wave: lanes l, l + 32 hold complementary rows
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = i * 32 + 4 * half + 8 * (r >> 2) + (r & 3);
                        s += rl < rows_here ? acc[j][i][r] : 0.f;
                    }
                s += __shfl_xor(s, 32);
                const float mean = s / (float)rows_here;
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = i * 32 + 4 * half + 8 * (r >> 2) + (r & 3);
                        const float d = acc[j][i][r] - mean;
                        m2 = fmaf(rl < rows_here ? d : 0.f, d, m2);
                    }
                m2 += __shfl_xor(m2, 32);
                if (half == 0) {
                    float* os = S.out_stats + (((long)b * ((M + PM - 1) / PM) + panel) * N + col) * 2;
                    os[0] = s;
                    os[1] = m2;
                }
            }
            const int pgrp = p.cpw > 0 ? col / p.cpw : 0, pcol = p.cpw > 0 ? col - pgrp * p.cpw : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = i * 32 + 4 * half + 8 * (r >> 2) + (r & 3);
                    if (rl < rows_here) {
                        const long grow = row0 + rl;
                        float v = acc[j][i][r];
                        if (R) v += R[grow * p.ldr + col];
                        if (C) C[grow * p.ldc + col] = v;
                        if (Cp) {
                            const _Float16 hi = (_Float16)v;
                            _Float16* dst = Cp + grow * p.ldcp + pgrp * 2 * p.cpw + pcol;
                            dst[0] = hi;
                            dst[p.cpw] = (_Float16)(v - (float)hi);
                        }
                    }
                }
        }
    };

    // ---- main loop ----------------------------------------------------------------------------------------------------------
    for (int q = 0; q < WST - 1; ++q) issue_w();           // the W ring starts before the A burst: both latencies overlap
    int pos = 0;
    zero_acc();
    for (int c = 0; c < nchunks; ++c) {
        if (c > 0 || ASRC == 1) __syncthreads();           // every wave is done with the previous A chunk / the norm table is there
        load_a_chunk(c);
        __syncthreads();                                   // (drains the A burst and whatever W stages are in flight)
        for (int g = 0; g < ngroups; ++g) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
#pragma unroll 2
                for (int ks = 0; ks < nks; ++ks) {
                    issue_w();                             // step pos + 3 into the stage consumed at pos - 1
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // 3 younger steps x 2 loads may still fly
                    const char* ws = wring + (pos & (WST - 1)) * WSTAGE_BYTES;
                    const char* as = Abuf + (ks >> 1) * 8192;
                    const int sx = (ks & 1) * 32;
                    f16x8 ah[2], al[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[i] = *reinterpret_cast<const f16x8*>(as + (aoff[i] ^ sx));
                        al[i] = *reinterpret_cast<const f16x8*>(as + 4096 + (aoff[i] ^ sx));
                    }
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(ws + woff);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(ws + 1024 + woff);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh, acc[j][i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl, acc[j][i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh, acc[j][i], 0, 0, 0);
                    ++pos;
                }
            }
            if (c == nchunks - 1) {
                epilogue(g);
                if (g + 1 < ngroups) zero_acc();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the wrapped-around prefetches must land before the LDS is released
}

template <int NG, int ASRC, int PRO>
hipError_t launch_panel(const PGemmParams& p, int batch, int panels_max, hipStream_t stream) {
    const int KC = p.K < KC_MAX ? p.K : KC_MAX;
    size_t lds = (size_t)(KC / BK) * 8192 + PW * WST * WSTAGE_BYTES;
    if (ASRC == 1) lds += (size_t)p.K * (PRO == 2 ? 4 : 2) * sizeof(float);
    if (hipError_t e = imp_grant_dynamic_lds((const void*)gemm_panel_kernel<NG, ASRC, PRO>, lds)) return e;
    hipLaunchKernelGGL((gemm_panel_kernel<NG, ASRC, PRO>), dim3(panels_max * p.nside * batch), dim3(512), lds, stream, p, panels_max);
    return hipGetLastError();
}

}  // namespace

// the panel kernel takes a GEMM when: planes or fp32-normalised A, no sub-batches / row vectors, N a multiple of 256 that is
// either <= 512 (all N accumulated in registers, any K = chunks of 256 or <= 256) or several 256-column groups with K <= 256,
// and enough panels to fill the chip
int pgemm_panel_groups(const PGemmParams& p, int batch, int* ng) {
    if (p.nsub != 1 || (p.flags & PG_EPI_EXPROW) || p.dbg) return 0;
    const int N = p.side[0].N, K = p.K;
    if (N % 256 || (K > KC_MAX && K % KC_MAX) || K % BK) return 0;
    if (K > KC_MAX && p.ksplit != K && p.ksplit != KC_MAX) return 0;
    if (K <= KC_MAX && p.ksplit != K) return 0;
    const int tiles = N / 256;
    int g = 0;
    if (tiles <= 2) g = tiles;
    else if (K <= KC_MAX) g = 1;
    else return 0;
    const int maxM = p.nside == 2 ? (p.side[0].M > p.side[1].M ? p.side[0].M : p.side[1].M) : p.side[0].M;
    const long panels = (long)((maxM + PM - 1) / PM) * p.nside * batch;
    if (panels < 96) return 0;
    *ng = g;
    return 1;
}

hipError_t launch_gemm_panel(const PGemmParams& p, int batch, int ng, hipStream_t stream) {
    const int maxM = p.nside == 2 ? (p.side[0].M > p.side[1].M ? p.side[0].M : p.side[1].M) : p.side[0].M;
    const int panels_max = (maxM + PM - 1) / PM;
    const bool a_f32 = p.side[0].Af != nullptr;
    const int pro = a_f32 ? ((!(p.flags & PG_PRO_AFFINE) && p.act == 0) ? 1 : 2) : 0;
#define IMP_PANEL(NGV)                                                                              \
    if (ng == NGV) {                                                                                \
        if (!a_f32) return launch_panel<NGV, 0, 0>(p, batch, panels_max, stream);                   \
        return pro == 1 ? launch_panel<NGV, 1, 1>(p, batch, panels_max, stream)                     \
                        : launch_panel<NGV, 1, 2>(p, batch, panels_max, stream);                    \
    }
    IMP_PANEL(1) IMP_PANEL(2)
#undef IMP_PANEL
    return hipErrorInvalidValue;
}
int pgemm_panel_rows() { return PM; }
