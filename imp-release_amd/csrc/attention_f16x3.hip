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
__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ int swap23(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

template <int DH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void attn_f16x3_kernel(const AttnParams p, int qtiles, int total_blocks) {
    constexpr int NT = NWAVES * 64;
    constexpr int KROW = DH + 4;                // floats per K row:  DH/2 (hi) + DH/2 (lo) + 4 pad
    constexpr int VROW = KT + 4;                // floats per V^T row: 32 (hi) + 32 (lo) + 4 pad
    constexpr int KF4 = KT * DH / 4, KLPT = KF4 / NT;          // float4 loads of K per thread
    constexpr int VG = KT * DH / 16, VGPT = (VG + NT - 1) / NT; // 16-key groups of V per thread
    constexpr int DT = DH / 32, KS = DH / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                           // [2][KT][KROW]
    float* Vs = Ks + 2 * KT * KROW;             // [2][DH][VROW]
    float* Bs = Vs + 2 * DH * VROW;             // [2][KT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int id = xcd_remap(blockIdx.x, total_blocks);
    const int qt = id % qtiles; id /= qtiles;
    const int h = id % IMP_NUM_HEADS; id /= IMP_NUM_HEADS;
    const int sidx = id % p.nside;
    const int b = id / p.nside;
    const AttnSide& S = p.side[sidx];
    const int nq = S.nq, nk = S.nk;
    const int q0 = qt * (NWAVES * 32);
    if (q0 >= nq) return;

    const float* Qg = S.q + b * S.sq_b + h * DH;
    const float* Kg = S.k + b * S.sk_b + h * DH;
    const float* Vg = S.v + b * S.sk_b + h * DH;
    const uint8_t* mk = S.kmask ? S.kmask + (long)b * nk : nullptr;

    // Q fragments: lane (query l31, half) holds d = 16 s + 8 half .. + 7 for k-step s, as hi / lo halves
    f16x8 qh[KS], ql[KS];
    {
        const int qrow = q0 + wave * 32 + l31;
        const float* src = Qg + (long)(qrow < nq ? qrow : nq - 1) * p.ldq + 8 * half;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + 16 * s);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + 16 * s + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hh, ll;
                split1(a[e], hh, ll); qh[s][e] = hh; ql[s][e] = ll;
                split1(c[e], hh, ll); qh[s][4 + e] = hh; ql[s][4 + e] = ll;
            }
        }
    }

    f32x4 rk[KLPT];
    float rv[VGPT][16];
    float rb = 0.f;
    // Staging loads go through raw buffer loads: SGPR descriptor (wave-uniform K / V base of this batch/head),
    // 32-bit per-lane byte offset (loop invariant), scalar byte offset for the tile / key row: no 64-bit VALU address
    // arithmetic in the loop, and rows past nk read as ZERO by the hardware bounds check (they are neutralised by the
    // -inf key bias anyway), so there is no clamped tail path.
    const unsigned kv_bytes = (unsigned)(((long)(nk - 1) * p.ldk + DH) * 4);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, kv_bytes, 0x00020000);
    int koff[KLPT], voff[VGPT];
#pragma unroll
    for (int j = 0; j < KLPT; ++j) {
        const int f = tid + j * NT;
        koff[j] = ((f / (DH / 4)) * p.ldk + (f % (DH / 4)) * 4) * 4;
    }
#pragma unroll
    for (int i = 0; i < VGPT; ++i) {
        const int g = tid + i * NT;
        voff[i] = (((g / DH) * 16) * p.ldk + (g % DH)) * 4;
    }
    const int row_bytes = p.ldk * 4;
    auto load_tile = [&](int t) {
        const int k0 = t * KT;
        const int soff = k0 * row_bytes;                    // uniform
#pragma unroll
        for (int j = 0; j < KLPT; ++j) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsK, koff[j], soff, 0);
            rk[j] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
#pragma unroll
        for (int i = 0; i < VGPT; ++i) {
            if (VG % NT == 0 || tid + i * NT < VG) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    rv[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsV, voff[i], soff + e * row_bytes, 0));
            }
        }
        if (tid < KT) {
            const int key = k0 + tid;
            bool ok = key < nk;
            if (ok && mk) ok = mk[key] != 0;
            rb = ok ? 0.f : -INFINITY;
        }
    };
    auto store_tile = [&](int buf) {
        float* ks = Ks + buf * KT * KROW;
        float* vs = Vs + buf * DH * VROW;
#pragma unroll
        for (int j = 0; j < KLPT; ++j) {
            const int f = tid + j * NT;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { _Float16 hh, ll; split1(rk[j][e], hh, ll); hi[e] = hh; lo[e] = ll; }
            *reinterpret_cast<f16x4*>(ks + row * KROW + (c4 >> 1)) = hi;
            *reinterpret_cast<f16x4*>(ks + row * KROW + DH / 2 + (c4 >> 1)) = lo;
        }
#pragma unroll
        for (int i = 0; i < VGPT; ++i) {
            const int g = tid + i * NT;
            if (VG % NT == 0 || g < VG) {
                const int d = g % DH, kg = g / DH;
                f16x8 hi[2], lo[2];
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) {          // position pos holds key swap23(pos) of this 16-key group
                    _Float16 hh, ll;
                    split1(rv[i][swap23(pos)], hh, ll);
                    hi[pos >> 3][pos & 7] = hh;
                    lo[pos >> 3][pos & 7] = ll;
                }
                float* row = vs + d * VROW + kg * 8;          // 16 halves = 8 floats per group
                *reinterpret_cast<f16x8*>(row) = hi[0];
                *reinterpret_cast<f16x8*>(row + 4) = hi[1];
                *reinterpret_cast<f16x8*>(row + KT / 2) = lo[0];
                *reinterpret_cast<f16x8*>(row + KT / 2 + 4) = lo[1];
            }
        }
        if (tid < KT) Bs[buf * KT + tid] = rb;
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float scale = DH == 64 ? 0.125f : 0.17677669529663687f;
    const float SL2E = scale * LOG2E;

    const int nt = (nk + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const float* ks = Ks + buf * KT * KROW + l31 * KROW + 4 * half;
        const float* vs = Vs + buf * DH * VROW + l31 * VROW + 4 * half;
        const float* bs = Bs + buf * KT;

        // ---- S^T = K . Q^T : 3 f16 MFMAs per 16-deep k-step, the two key blocks interleaved ------------------
        f32x16 sacc[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[jb][r] = 0.f;
        {
            f16x8 kh[2][KS], kl[2][KS];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    kh[jb][s] = *reinterpret_cast<const f16x8*>(ks + jb * 32 * KROW + 8 * s);
                    kl[jb][s] = *reinterpret_cast<const f16x8*>(ks + jb * 32 * KROW + DH / 2 + 8 * s);
                }
            // product-major order: consecutive MFMAs alternate between the two accumulators (no dependent pairs)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[0][s], qh[s], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[1][s], qh[s], sacc[1], 0, 0, 0);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[0][s], ql[s], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[1][s], ql[s], sacc[1], 0, 0, 0);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[0][s], qh[s], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[1][s], qh[s], sacc[1], 0, 0, 0);
            }
        }
        // ---- online softmax on the raw dot products (scale folded into the exp2 constant) ---------------------
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
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        // ---- P = exp(S - m) (split to hi/lo halves in registers) and O^T += V^T . P^T -------------------------
        float lsum = 0.f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 vh[DT], vl[DT];
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    vh[d] = *reinterpret_cast<const f16x8*>(vs + d * 32 * VROW + jb * 16 + 8 * s2);
                    vl[d] = *reinterpret_cast<const f16x8*>(vs + d * 32 * VROW + KT / 2 + jb * 16 + 8 * s2);
                }
                f16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = fast_exp2(fmaf(sacc[jb][8 * s2 + e], SL2E, mneg));
                    lsum += pv;
                    _Float16 hh, ll;
                    split1(pv, hh, ll);
                    ph[e] = hh; pl[e] = ll;
                }
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[d], ph, oacc[d], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[d], pl, oacc[d], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[d], ph, oacc[d], 0, 0, 0);
            }
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise, transpose through LDS, coalesced row stores ------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    constexpr int LDO = DH + 1;
    float* ot = smem + wave * 32 * LDO;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ot[l31 * LDO + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = oacc[d][r] / l_tot;
    if (S.lse && half == 0) {
        const int qrow = q0 + wave * 32 + l31;
        if (qrow < nq) S.lse[((long)b * IMP_NUM_HEADS + h) * nq + qrow] = m_run + logf(l_tot);
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
}

template <int DH, int NWAVES>
hipError_t launch_one(const AttnParams& p, int batch, int maxq, hipStream_t stream) {
    const int qtiles = (maxq + NWAVES * 32 - 1) / (NWAVES * 32);
    const int total = qtiles * IMP_NUM_HEADS * p.nside * batch;
    const size_t lds = (size_t)(2 * KT * (DH + 4) + 2 * DH * (KT + 4) + 2 * KT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_f16x3_kernel<DH, NWAVES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_f16x3_kernel<DH, NWAVES>), dim3(total), dim3(NWAVES * 64), lds, stream, p, qtiles, total);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_f16x3(const AttnParams& p, int batch, hipStream_t stream) {
    int maxq = p.side[0].nq;
    if (p.nside == 2 && p.side[1].nq > maxq) maxq = p.side[1].nq;
    if (maxq <= 0 || batch <= 0) return hipSuccess;
    const long wg4 = (long)((maxq + 127) / 128) * IMP_NUM_HEADS * p.nside * batch;
    const bool big = wg4 >= 256;
    // 8-wave workgroups (256 queries) when that still gives >= 1 workgroup per CU: every K/V tile is staged and split
    // once per 256 queries instead of once per 128 (measured 127 -> 114 us at N=2048, B=4)
    if (p.dh == 64 && (long)((maxq + 255) / 256) * IMP_NUM_HEADS * p.nside * batch >= 256)
        return launch_one<64, 8>(p, batch, maxq, stream);
    if (p.dh == 64) return big ? launch_one<64, 4>(p, batch, maxq, stream) : launch_one<64, 2>(p, batch, maxq, stream);
    if (p.dh == 32) return big ? launch_one<32, 4>(p, batch, maxq, stream) : launch_one<32, 2>(p, batch, maxq, stream);
    return hipErrorInvalidValue;
}
