// Multi-head attention core (nets/layers.py:121-131) for gfx950, fp32 end to end on the matrix pipe
// (v_mfma_f32_32x32x2_f32), flash-style: the [B][4][N][M] probability tensor that the reference
// materialises and keeps (nets/layers.py:132) is never written; per query row only the running
// max / sum and - optionally - the log-sum-exp are kept (enough to re-create any probability later).
//
// Work decomposition: one workgroup = NWAVES wave64, each wave owns 32 queries of one
// (batch, image side, head); keys/values stream through LDS in tiles of 64 keys, double-buffered
// (global -> registers -> LDS, one barrier per tile).
//
// "Swapped" QK^T: the wave computes S^T = K.Q^T (A operand = K rows from LDS, B operand = Q held in
// registers), so in the MFMA C layout the query is the lane (column) and the 32 keys of a block are
// spread over 16 registers x 2 lane halves.  Then
//   * the softmax row reduction is in-register + ONE cross-half exchange,
//   * the probabilities are already in the B-operand layout of the PV product
//     O^T[d][q] += V^T[d][key] . P^T[key][q]   (register r holds keys kappa(r) | kappa(r)+4 in the two
//     lane halves = exactly one k-pair of a 32x32x2 MFMA), so P never leaves registers.
// The running rescale factor is per query = per lane.
#include "imp_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KT = 64;                 // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// bijective XCD-aware remap of a linear block id: consecutive logical ids (which share K/V) run on one XCD
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = lin % 8, idx = lin / 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int DH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, (NWAVES == 4 ? 2 : 1)) void attn_f32_kernel(const AttnParams p, int qtiles, int total_blocks) {
    constexpr int NT = NWAVES * 64;
    constexpr int LDK = DH + 4;                 // padded K row: 16 consecutive rows -> 16 distinct 16B slots
    constexpr int F4 = KT * DH / 4;             // float4 per K (or V) tile
    constexpr int LPT = F4 / NT;                // float4 loads per thread per operand
    constexpr int DT = DH / 32;                 // 32-wide output tiles over the head dim
    constexpr int QS = DH / 2;                  // MFMA k-steps of QK^T
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                           // [2][KT][LDK]
    float* Vs = Ks + 2 * KT * LDK;              // [2][KT][DH]
    float* Bs = Vs + 2 * KT * DH;               // [2][KT] additive key bias (0 or -inf)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int id = xcd_remap(blockIdx.x, total_blocks);
    const int qt = id % qtiles; id /= qtiles;
    const int h = id % IMP_NUM_HEADS; id /= IMP_NUM_HEADS;
    const int sidx = id % p.nside;
    const int b = id / p.nside;
    const AttnSide& S = p.side[sidx];
    const int nq = imp_count(p.rc, S.qimg, b, S.nq), nk = imp_count(p.rc, S.kimg, b, S.nk);      // ragged batches: this pair's own counts; S.nq / S.nk = the padded layout
    const int q0 = qt * (NWAVES * 32);
    if (q0 >= nq || nk <= 0) return;
    const bool clk_on = p.clk_probe != nullptr && blockIdx.x == 0;      // timing hook (imp_kernels.h AttnParams::clk_probe)
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }

    const float* Qg = S.q + b * S.sq_b + h * DH;
    const float* Kg = S.k + b * S.sk_b + h * DH;
    const float* Vg = S.v + b * S.sk_b + h * DH;
    const uint8_t* mk = S.kmask ? S.kmask + (long)b * S.nk : nullptr;

    // Q fragment: lane (query l31, half) holds Q[q][half*QS + s], s = 0..QS-1
    float qreg[QS];
    {
        const int qrow = q0 + wave * 32 + l31;
        const float* src = Qg + (long)(qrow < nq ? qrow : nq - 1) * p.ldq + half * QS;
#pragma unroll
        for (int c = 0; c < QS / 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * c);
            qreg[4 * c + 0] = v[0]; qreg[4 * c + 1] = v[1]; qreg[4 * c + 2] = v[2]; qreg[4 * c + 3] = v[3];
        }
    }

    f32x4 rk[LPT], rv[LPT];
    float rb = 0.f;
    auto load_tile = [&](int t) {
        const int k0 = t * KT;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int f = tid + j * NT;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            // tail keys are clamped to the last valid row (finite data) and neutralised by the -inf key bias
            const long krow = (long)min(k0 + row, nk - 1) * p.ldk + c4;
            rk[j] = *reinterpret_cast<const f32x4*>(Kg + krow);
            rv[j] = *reinterpret_cast<const f32x4*>(Vg + krow);
        }
        if (tid < KT) {
            const int key = k0 + tid;
            bool ok = key < nk;
            if (ok && mk) ok = mk[key] != 0;
            rb = ok ? 0.f : -INFINITY;
        }
    };
    auto store_tile = [&](int buf) {
        float* ks = Ks + buf * KT * LDK;
        float* vs = Vs + buf * KT * DH;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int f = tid + j * NT;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            *reinterpret_cast<f32x4*>(ks + row * LDK + c4) = rk[j];
            *reinterpret_cast<f32x4*>(vs + row * DH + c4) = rv[j];
        }
        if (tid < KT) Bs[buf * KT + tid] = rb;
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float scale = DH == 64 ? 0.125f : 0.17677669529663687f;   // 1/sqrt(DH) (exact for DH = 64)
    const float SL2E = scale * LOG2E;

    const int nt = (nk + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const float* ks = Ks + buf * KT * LDK;
        const float* vs = Vs + buf * KT * DH;
        const float* bs = Bs + buf * KT;

        // ---- S^T = K . Q^T for the two 32-key blocks of the tile: the two accumulation chains are interleaved
        //      and the K fragments of step c+1 are read while step c is multiplied (no LDS wait inside a chain)
        f32x16 sacc[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[jb][r] = 0.f;
        {
            const float* krow0 = ks + l31 * LDK + half * QS;
            const float* krow1 = krow0 + 32 * LDK;
            f32x4 kf[2][2];
            kf[0][0] = *reinterpret_cast<const f32x4*>(krow0);
            kf[0][1] = *reinterpret_cast<const f32x4*>(krow1);
#pragma unroll
            for (int c = 0; c < QS / 4; ++c) {
                if (c + 1 < QS / 4) {
                    kf[(c + 1) & 1][0] = *reinterpret_cast<const f32x4*>(krow0 + 4 * (c + 1));
                    kf[(c + 1) & 1][1] = *reinterpret_cast<const f32x4*>(krow1 + 4 * (c + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c & 1][0][e], qreg[4 * c + e], sacc[0], 0, 0, 0);
                    sacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c & 1][1][e], qreg[4 * c + e], sacc[1], 0, 0, 0);
                }
            }
        }
        // ---- online softmax on the RAW dot products: s = raw * scale; scale (> 0) is folded into the exp2 constant and
        //      into the running max.  Only a tile with invalid keys (tail of the key range, or a key mask) pays for the
        //      additive -inf key bias (uniform branch) ---------------------------------------------------------------
        if (mk != nullptr || (t + 1) * KT > nk) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + jb * 32 + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sacc[jb][4 * g + e] += bias[e];
                }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[jb][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32)) * scale;
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2((m_run - m_use) * LOG2E);
        const float mneg = -m_use * LOG2E;
        if (__any(alpha != 1.f)) {          // wave-uniform: the running max rarely moves after the first tiles
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        // ---- P = exp(S - m) and O^T += V^T . P^T, software-pipelined over 8 groups of 4 keys:
        //      iteration g issues the LDS reads of group g+1's V operands, exponentiates group g+1's scores (VALU)
        //      and multiplies group g (MFMA).  sched_barrier pins "reads first" so no MFMA waits on an LDS round trip.
        float lsum = 0.f;
        {
            float vv[2][4][DT];
            const float* vbase = vs + (4 * half) * DH + l31;
            auto loadv = [&](int buf2, int idx) {          // idx = jb*4 + g ; keys jb*32 + 8g + rr (+4 for the upper half)
                const float* vp = vbase + ((idx >> 2) * 32 + (idx & 3) * 8) * DH;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int d = 0; d < DT; ++d) vv[buf2][rr][d] = vp[rr * DH + d * 32];
            };
            auto expg = [&](int idx) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float pv = fast_exp2(fmaf(sacc[idx >> 2][(idx & 3) * 4 + rr], SL2E, mneg));
                    sacc[idx >> 2][(idx & 3) * 4 + rr] = pv;
                    lsum += pv;
                }
            };
            loadv(0, 0);
            expg(0);
#pragma unroll
            for (int idx = 0; idx < 8; ++idx) {
                if (idx + 1 < 8) loadv((idx + 1) & 1, idx + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (idx + 1 < 8) expg(idx + 1);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int d = 0; d < DT; ++d)
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[idx & 1][rr][d], sacc[idx >> 2][(idx & 3) * 4 + rr],
                                                                       oacc[d], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise, transpose through LDS, coalesced row stores -------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    constexpr int LDO = DH + 1;
    float* ot = smem + wave * 32 * LDO;        // K/V tiles are dead after the loop's last barrier
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ot[l31 * LDO + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = imp_div_by(oacc[d][r], l_tot, inv_l);        // == oacc / l_tot, bit for bit
    if (S.lse && half == 0) {
        const int qrow = q0 + wave * 32 + l31;
        if (qrow < nq) S.lse[((long)b * IMP_NUM_HEADS + h) * S.nq + qrow] = m_run + logf(l_tot);
    }
    __syncthreads();
    float* Og = S.out + b * S.so_b + h * DH;
    if (DH == 64) {
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int qrow = q0 + wave * 32 + i;
            if (qrow < nq) Og[(long)qrow * p.ldo + lane] = ot[i * LDO + lane];
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int qi = 2 * i + half;
            const int qrow = q0 + wave * 32 + qi;
            if (qrow < nq) Og[(long)qrow * p.ldo + l31] = ot[qi * LDO + l31];
        }
    }
    if (clk_on && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        p.clk_probe[0] += __builtin_readcyclecounter() - clk_c0;
        p.clk_probe[1] += __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}

template <int DH, int NWAVES>
hipError_t launch_one(const AttnParams& p, int batch, int maxq, hipStream_t stream) {
    const int qtiles = (maxq + NWAVES * 32 - 1) / (NWAVES * 32);
    const int total = qtiles * IMP_NUM_HEADS * p.nside * batch;
    const size_t lds = (size_t)(2 * KT * (DH + 4) + 2 * KT * DH + 2 * KT) * sizeof(float);
    if (hipError_t e = imp_grant_dynamic_lds((const void*)attn_f32_kernel<DH, NWAVES>, lds)) return e;
    hipLaunchKernelGGL((attn_f32_kernel<DH, NWAVES>), dim3(total), dim3(NWAVES * 64), lds, stream, p, qtiles, total);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_f32(const AttnParams& p, int batch, hipStream_t stream) {
    int maxq = p.side[0].nq;
    if (p.nside == 2 && p.side[1].nq > maxq) maxq = p.side[1].nq;
    if (maxq <= 0 || batch <= 0) return hipSuccess;
    // 4-wave workgroups (128 queries) when that still gives >= ~1 workgroup per CU, else 2-wave (64 queries)
    const long wg4 = (long)((maxq + 127) / 128) * IMP_NUM_HEADS * p.nside * batch;
    const bool big = wg4 >= 256;
    if (p.dh == 64) return big ? launch_one<64, 4>(p, batch, maxq, stream) : launch_one<64, 2>(p, batch, maxq, stream);
    if (p.dh == 32) return big ? launch_one<32, 4>(p, batch, maxq, stream) : launch_one<32, 2>(p, batch, maxq, stream);
    return hipErrorInvalidValue;
}
