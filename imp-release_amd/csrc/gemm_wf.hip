// Weight-fragment GEMM for the three 1x1 convolutions of a GNN layer (nets/layers.py:119-120 q|k|v, :145-149 / :210-218 the MLP):
//
//   C[M x N] = epilogue( prologue(A)[M x K] . W[N x K]^T ),   K = 256 or 512, N a multiple of 128, W static
//
// What is different from gemm_f32.hip (which converts and stages BOTH operands per K-tile behind two barriers): the weights are
// split into f16 hi / lo halves and re-ordered into MFMA fragment order ONCE, when they are loaded (`wf_pack`), so a wave
// fetches a fragment as one coalesced 1-KB load from L2 and the weights never touch LDS; the activations of a 64-row tile are
// converted ONCE into hi / lo half planes covering the whole K extent (67.6 KB for K = 256: 2 workgroups per CU) and every
// column pass of the tile re-reads them from LDS.  Between the staging barrier and the end of the workgroup there is no barrier:
// 4 waves, each 64 rows x 32 columns of a 128-column pass, K loop software-pipelined by hand exactly like csrc/superpoint.hip
// (A fragments one k-step ahead, B fragments four k-steps ahead).  Arithmetic: split-half f16x3, products lo.hi + hi.lo + hi.hi
// in that order, fp32 accumulate - the scheme of gemm_f32.hip (PREC = 1).
//
// LDS plane: row pitch = 2 K + 16 bytes = 1 (mod 16) sixteen-byte slots, so the 16 rows of a ds_read_b128 lane group
// ({0-3,12-15,20-27} ...) fall into 16 distinct slots.
//
// Two epilogue shapes: SWAP = 1 (weights as first MFMA operand: a register holds 4 consecutive columns of the lane's row, 16-byte
// stores, bias and residual as 16-byte loads) and SWAP = 0 (a lane holds one column of 32 rows: the per-block InstanceNorm
// statistics (sum, M2 about the block mean) of the MLP's first convolution are register reductions, as in gemm_f32.hip).
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

#ifdef WF_PROFILE   // tools/build_variant.sh -DWF_PROFILE: cycle stamps of wave 0 of every workgroup: staging, K loops, epilogues, total
__device__ unsigned long long wf_prof[4096][4];
#endif

template <int K, int PRO, int SWAP, int STATS>
__global__ __launch_bounds__(256) void gemm_wf_kernel(const WfParams p, int row_tiles) {
    constexpr int PITCH = 2 * K + 16;               // bytes per row of one half plane
    constexpr int PLANE = WF_TM * PITCH;
    constexpr int NS = K / 16;                      // k-steps
    extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    int z = blockIdx.x;
    // small launches: the column passes of a tile are dealt to `psplit` workgroups (each stages the tile itself) to fill the chip
    const int psplit = p.pass_split > 1 ? p.pass_split : 1;
    const int pgrp = z % psplit; z /= psplit;
    const int rtile = z % row_tiles; z /= row_tiles;
    const int sidx = z % p.nside;
    const int b = z / p.nside;
    const WfSide& S = p.side[sidx];
    const int M = S.M, N = p.N;
    const int row0 = rtile * WF_TM;
    if (row0 >= M) return;                          // uniform per workgroup, before any barrier
#ifdef WF_PROFILE
    const unsigned long long t_begin = __builtin_readcyclecounter();
    unsigned long long t_k = 0, t_e = 0, t_x;
#endif

    // ---- PRO: the InstanceNorm constants (mean, rstd) of this batch element's K channels -> LDS.  Either finalised by
    // stats_finalize_kernel (in_stats) or merged here from the producer's per-block (sum, M2) with Chan's formula in fp64, in the
    // arithmetic and summation order of that kernel (4 block groups t = g, g + 4, ...; ((g0 + g1) + (g2 + g3)))
    float* stl = reinterpret_cast<float*>(wf_smem + 2 * PLANE + 4 * 32 * 144);     // [K][2]
    if (PRO) {
        if (S.stat_part) {
            const int T = S.stat_tiles, TR = p.stat_tile_rows;
            for (int k = tid; k < K; k += 256) {
                const float2* sp = reinterpret_cast<const float2*>(S.stat_part) + (long)b * T * K + k;
                double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, a3[4] = {0, 0, 0, 0};
                for (int t = 0; t < T; ++t) {
                    const float2 v = sp[(long)t * K];
                    const int nt = min(TR, M - t * TR);
                    a1[t & 3] += (double)v.x; a2[t & 3] += (double)v.y; a3[t & 3] += (double)v.x * (double)v.x / (double)nt;
                }
                const double s1t = (a1[0] + a1[1]) + (a1[2] + a1[3]);
                const double m2w = (a2[0] + a2[1]) + (a2[2] + a2[3]);
                const double sqn = (a3[0] + a3[1]) + (a3[2] + a3[3]);
                const double mean = s1t / (double)M;
                double m2 = m2w + (sqn - (double)M * mean * mean);
                m2 = m2 < 0.0 ? 0.0 : m2;
                stl[2 * k] = (float)mean;
                stl[2 * k + 1] = (float)(1.0 / sqrt(m2 / (double)M + (double)p.norm_eps));
            }
        } else {
            const float* st = S.in_stats + (long)b * K * 2;
            for (int i = tid; i < 2 * K; i += 256) stl[i] = st[i];
        }
        __syncthreads();
    }
    // ---- stage the tile: 64 rows x K fp32 -> hi / lo half planes (rows past M are clamped: they only feed rows that are never stored)
    {
        const float* A = S.A + b * S.sA_b;
        const float* A2 = S.A2 ? S.A2 + b * S.sA2_b : A;
        const float* st = stl;
        constexpr int F4_ROW = K / 4;               // float4 per row
        constexpr int PER = WF_TM * F4_ROW / 256;   // float4 per thread
        f32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = j * 256 + tid;
            const int r = e / F4_ROW, k = (e - r * F4_ROW) * 4;
            const int gr = min(row0 + r, M - 1);
            const float* src = k < p.ksplit ? A + (long)gr * p.lda + k : A2 + (long)gr * p.lda2 + (k - p.ksplit);
            v[j] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = j * 256 + tid;
            const int r = e / F4_ROW, k = (e - r * F4_ROW) * 4;
            f32x4 x = v[j];
            if (PRO) {                              // InstanceNorm with the producer's finalised statistics + ReLU (nets/layers.py:67-76)
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(st + 2 * k), s1 = *reinterpret_cast<const f32x4*>(st + 2 * k + 4);
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
#ifdef WF_PROFILE
    const unsigned long long t_staged = __builtin_readcyclecounter();
#endif

    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = (32 * i + (lane & 31)) * PITCH + half * 16;
    const int npass_all = N >> 7;
    const int npass = npass_all / psplit;          // passes of this workgroup: pbase .. pbase + npass
    const int pbase = pgrp * npass;
    const u32x4* wbase = reinterpret_cast<const u32x4*>(p.Wf_) + lane;
    auto wptr = [&](int pass) { return wbase + (size_t)(pass * 4 + wave) * NS * 128; };

    // workgroups start at different column passes and wrap around: they all begin at the same time, and walking the weights in
    // the same order would have every CU ask the L2 for the same lines at the same moment
    const int pass0 = (int)((blockIdx.x / psplit) % (unsigned)npass);
    u32x4 bh[4], bl[4];
    {
        const u32x4* w0 = wptr(pbase + pass0);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bh[c] = w0[c * 128]; bl[c] = w0[c * 128 + 64]; }
    }
    // Epilogue through a wave-private LDS transposition (32 rows x 32 columns at a time, row pitch 144 bytes): whatever the
    // accumulator layout, the tile is read back row-major - lane = 4 consecutive columns of one of 8 rows - so bias and residual
    // are coalesced 16-byte loads and every store instruction writes 8 FULL 128-byte lines.  (Measured before, storing straight
    // from the accumulator layout - 32 lines x 32 bytes per instruction -: 1.8 k cycles per pass against 3.3 k of K loop.)
    constexpr int TP = 144;
    unsigned char* const tbuf = wf_smem + 2 * PLANE + wave * (32 * TP);
    float* const Cb = S.C + b * S.sC_b;
    const float* const Rb = S.R ? S.R + b * S.sR_b : nullptr;
#pragma unroll 1
    for (int pi = 0; pi < npass; ++pi) {
        const int pl = pass0 + pi < npass ? pass0 + pi : pass0 + pi - npass;
        const int pass = pbase + pl;
        const int pnext = pbase + (pl + 1 < npass ? pl + 1 : 0);
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const u32x4* wp = wptr(pass);
        const u32x4* wfollow = wptr(pnext);                                   // first group of the next pass (after the last: a harmless load)
        f16x8 fah[2][2], fal[2][2];
        auto load_frag = [&](int st, int fb) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fah[fb][i] = *reinterpret_cast<const f16x8*>(wf_smem + aoff[i] + st * 32);
                fal[fb][i] = *reinterpret_cast<const f16x8*>(wf_smem + PLANE + aoff[i] + st * 32);
            }
        };
#ifdef WF_PROFILE
        t_x = __builtin_readcyclecounter();
#endif
        // the residual of this pass in the read-back layout of the epilogue, requested now: it arrives under the K loop
        f32x4 rres[2][4];
        if (Rb && !(p.dbg & 4)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = min(row0 + 32 * i + 8 * j + (lane >> 3), M - 1);
                    rres[i][j] = *reinterpret_cast<const f32x4*>(Rb + (long)row * p.ldr + pass * 128 + wave * 32 + (lane & 7) * 4);
                }
        }
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
#ifdef WF_PROFILE
        { const unsigned long long t = __builtin_readcyclecounter(); t_k += t - t_x; t_x = t; }
#endif
        const int cb = pass * 128 + wave * 32;                   // first column of this wave in this pass
        if (STATS) {
            // SWAP = 0: register r of fragment i = row 32 i + 4 half + (r & 3) + 8 (r >> 2) of column cb + lane % 32.
            // (sum, M2 about the block mean) of this 64-row block per column, on the values as stored (bias included)
            const int col = cb + (lane & 31);
            const float bvs = p.bias ? p.bias[col] : 0.f;
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
            if (half == 0) {
                const int tiles_side = (M + WF_TM - 1) / WF_TM;            // the buffer is [b][this side's blocks][N][2] (stats_finalize_kernel)
                float* o = S.out_stats + (((long)b * tiles_side + rtile) * N + col) * 2;
                o[0] = sum;
                o[1] = m2;
            }
        }
        if (p.dbg & 2) { if (acc[0][0] == 123.456f) Cb[0] = acc[1][5]; continue; }
        const int c4 = (lane & 7) * 4;
        const f32x4 bias4 = (p.bias && !(p.dbg & 4)) ? *reinterpret_cast<const f32x4*>(p.bias + cb + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (SWAP) {                                          // register r = column 4 half + (r & 3) + 8 (r >> 2) of row lane % 32
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e];
                    *reinterpret_cast<f32x4*>(tbuf + (lane & 31) * TP + (4 * half + 8 * g) * 4) = v;
                }
            } else {                                             // register r = row 4 half + (r & 3) + 8 (r >> 2) of column lane % 32
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(tbuf + (4 * half + (r & 3) + 8 * (r >> 2)) * TP + (lane & 31) * 4) = acc[i][r];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lr = 8 * j + (lane >> 3);
                const int row = row0 + 32 * i + lr;
                f32x4 v = *reinterpret_cast<const f32x4*>(tbuf + lr * TP + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bias4[e];
                if (row < M) {
                    if (Rb && !(p.dbg & 4)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rres[i][j][e];
                    }
                    if (!(p.dbg & 1) || v[0] == 123.456f) *reinterpret_cast<f32x4*>(Cb + (long)row * p.ldc + cb + c4) = v;
                }
            }
        }
#ifdef WF_PROFILE
        t_e += __builtin_readcyclecounter() - t_x;
#endif
    }
#ifdef WF_PROFILE
    if (tid == 0 && blockIdx.x < 4096) {
        wf_prof[blockIdx.x][0] = t_staged - t_begin; wf_prof[blockIdx.x][1] = t_k; wf_prof[blockIdx.x][2] = t_e;
        wf_prof[blockIdx.x][3] = __builtin_readcyclecounter() - t_begin;
    }
#endif
}

template <int K, int PRO, int SWAP, int STATS>
hipError_t wf_launch(const WfParams& p, int batch, hipStream_t stream) {
    int maxm = p.side[0].M;
    if (p.nside > 1 && p.side[1].M > maxm) maxm = p.side[1].M;
    const int row_tiles = (maxm + WF_TM - 1) / WF_TM;
    constexpr size_t lds = (size_t)2 * WF_TM * (2 * K + 16) + 4 * 32 * 144 + (PRO ? 2 * K * 4 : 0);   // half planes + transposition buffers + norm constants
    if (hipError_t e = imp_grant_dynamic_lds((const void*)gemm_wf_kernel<K, PRO, SWAP, STATS>, lds)) return e;
    const int psplit = p.pass_split > 1 ? p.pass_split : 1;
    if ((p.N >> 7) % psplit) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_wf_kernel<K, PRO, SWAP, STATS>), dim3(batch * p.nside * row_tiles * psplit), dim3(256), lds, stream, p, row_tiles);
#ifdef WF_PROFILE
    {
        static int calls = 0;
        if (++calls % 21 == 0) {
            (void)hipStreamSynchronize(stream);
            const int nb = batch * p.nside * row_tiles * psplit < 4096 ? batch * p.nside * row_tiles * psplit : 4096;
            std::vector<unsigned long long> h((size_t)nb * 4);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(wf_prof), h.size() * 8);
            double a[4] = {0, 0, 0, 0};
            for (int i = 0; i < nb; ++i) for (int j = 0; j < 4; ++j) a[j] += (double)h[(size_t)i * 4 + j] / nb;
            fprintf(stderr, "[gemm_wf<%d,%d,%d,%d> N=%d, %d workgroups] mean cycles: staging %.0f  K loops %.0f  epilogues %.0f  total %.0f\n", K, PRO, SWAP, STATS, p.N, nb,
                    a[0], a[1], a[2], a[3]);
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
    const bool pro = p.side[0].in_stats != nullptr || p.side[0].stat_part != nullptr;
    if (!gemm_wf_supported(p.K, p.N)) return hipErrorInvalidValue;
    if (stats) {
        if (pro) return hipErrorInvalidValue;
        return p.K == 256 ? wf_launch<256, 0, 0, 1>(p, batch, stream) : wf_launch<512, 0, 0, 1>(p, batch, stream);
    }
    if (pro) return p.K == 256 ? wf_launch<256, 1, 1, 0>(p, batch, stream) : wf_launch<512, 1, 1, 0>(p, batch, stream);
    return p.K == 256 ? wf_launch<256, 0, 1, 0>(p, batch, stream) : wf_launch<512, 0, 1, 0>(p, batch, stream);
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
