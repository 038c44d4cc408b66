// Small kernels around the GEMM / attention / OT cores (gfx950):
//   normalize_keypoints                nets/layers.py:49-56
//   KeypointEncoder first 3->C0 conv   nets/layers.py:85,88-90   (K = 3 is not an MFMA shape: VALU)
//   attention column sums for pooling  nets/adgm.py:557-565      (MFMA, roles of Q and K swapped)
//   AdaGMN.pool selection              nets/adgm.py:567-605      (threshold, LOWER medians by radix select,
//                                                                 union, wave-ballot stream compaction)
//   ragged gather                      eval/matching.py:166-174
#include "imp_kernels.h"
#include <map>
#include <mutex>
#include <utility>

hipError_t imp_grant_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& g = granted[std::make_pair(dev, kernel)];
    if (bytes > g) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        g = bytes;
    }
    return hipSuccess;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ kpts, long count, float cx, float cy,
                                                        float scaling, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float x = kpts[2 * i], y = kpts[2 * i + 1];
    out[2 * i] = (x - cx) / scaling;
    out[2 * i + 1] = (y - cy) / scaling;
}

// one thread per keypoint, 128 keypoints per block; every WAVE (64 keypoints) is one statistics block for the following
// InstanceNorm: (sum, M2 about the block mean) per channel, merged by launch_stats_finalize (Chan's formula)
__global__ __launch_bounds__(128) void kenc_first_kernel(Kenc0Side s0, Kenc0Side s1, int c0,
                                                         const float* __restrict__ W0, const float* __restrict__ b0,
                                                         float cx, float cy, float scaling, RaggedCounts rc) {
    const Kenc0Side& S = blockIdx.y == 0 ? s0 : s1;
    const int b = blockIdx.z, npad = S.n;                       // strides and the statistics layout follow the padded count
    const int n = imp_count(rc, blockIdx.y, b, npad);           // ragged batches: this pair's own keypoint count
    if ((int)blockIdx.x * 128 >= n) return;
    const int tok = blockIdx.x * 128 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool ok = tok < n;
    float x = 0.f, y = 0.f, sc = 0.f;
    if (ok) {
        x = S.kpts[((long)b * npad + tok) * 2];
        y = S.kpts[((long)b * npad + tok) * 2 + 1];
        sc = S.scores[(long)b * npad + tok];
        if (scaling > 0.f) { x = (x - cx) / scaling; y = (y - cy) / scaling; }
    }
    const int blk = blockIdx.x * 2 + wave, nblk = (npad + 63) / 64;
    const int cnt = min(64, n - blk * 64);                      // valid keypoints of this wave's block
    for (int c = 0; c < c0; ++c) {
        float v = fmaf(W0[c * 3 + 2], sc, fmaf(W0[c * 3 + 1], y, W0[c * 3] * x)) + b0[c];
        if (ok) S.y[((long)b * npad + tok) * c0 + c] = v; else v = 0.f;
        if (S.stats && cnt > 0) {                               // wave-uniform
            const float s = wave_sum(v);
            const float d = v - s / (float)cnt;
            const float m2 = wave_sum(ok ? d * d : 0.f);
            if (lane == 0) {
                float* st = S.stats + (((long)b * nblk + blk) * c0 + c) * 2;
                st[0] = s;
                st[1] = m2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// colsum[b][h][key] = sum_q exp(q.k / sqrt(dh) - lse[b][h][q])   : attention mass a key receives.
// S = Q.K^T with A = Q rows streamed through LDS (i = query) and B = K held in registers (j = key = lane),
// so the sum over queries is an in-lane sum over the 16 accumulator registers (+ one cross-half add).
constexpr int QT = 64;
constexpr float LOG2E = 1.4426950408889634f;

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_colsum_kernel(const ColsumParams p, int ktiles) {
    constexpr int LDQ = DH + 4, QS = DH / 2, F4 = QT * DH / 4, LPT = F4 / 256;
    __shared__ __attribute__((aligned(16))) float Qs[2][QT][LDQ];
    __shared__ __attribute__((aligned(16))) float Ls[2][QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int id = blockIdx.x;
    const int kt = id % ktiles; id /= ktiles;
    const int h = id % IMP_NUM_HEADS; id /= IMP_NUM_HEADS;
    const int sidx = id % p.nside;
    const int b = id / p.nside;
    const ColsumSide& S = p.side[sidx];
    const int nq = S.nq, nk = S.nk;
    const int k0 = kt * 128;
    if (k0 >= nk) return;
    const float* Qg = S.q + b * S.sq_b + h * DH;
    const float* Kg = S.k + b * S.sk_b + h * DH;
    const float* Lg = S.lse + ((long)b * IMP_NUM_HEADS + h) * (S.lq ? S.lq : nq);

    float kreg[QS];
    {
        const int krow = k0 + wave * 32 + l31;
        const float* src = Kg + (long)(krow < nk ? krow : nk - 1) * p.ldk + half * QS;
#pragma unroll
        for (int c = 0; c < QS / 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * c);
            kreg[4 * c] = v[0]; kreg[4 * c + 1] = v[1]; kreg[4 * c + 2] = v[2]; kreg[4 * c + 3] = v[3];
        }
    }
    f32x4 rq[LPT];
    float rl = 0.f;
    auto load_tile = [&](int t) {
        const int q0 = t * QT;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int f = tid + j * 256;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            rq[j] = *reinterpret_cast<const f32x4*>(Qg + (long)min(q0 + row, nq - 1) * p.ldq + c4);   // lse = +inf masks the tail
        }
        if (tid < QT) rl = (q0 + tid < nq) ? Lg[q0 + tid] : INFINITY;   // exp(s - inf) = 0 for padded queries
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int f = tid + j * 256;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            *reinterpret_cast<f32x4*>(&Qs[buf][row][c4]) = rq[j];
        }
        if (tid < QT) Ls[buf][tid] = rl;
    };
    float colacc = 0.f;
    const int nt = (nq + QT - 1) / QT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const float* qrow = &Qs[buf][jb * 32 + l31][half * QS];
#pragma unroll
            for (int c = 0; c < QS / 4; ++c) {
                const f32x4 qf = *reinterpret_cast<const f32x4*>(qrow + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[e], kreg[4 * c + e], sacc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(&Ls[buf][jb * 32 + 8 * g + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = DH == 64 ? sacc[4 * g + e] * 0.125f : sacc[4 * g + e] / 5.656854249492381f;
                    colacc += __builtin_amdgcn_exp2f((s - ls[e]) * LOG2E);
                }
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }
    colacc += __shfl_xor(colacc, 32);
    const int key = k0 + wave * 32 + l31;
    if (half == 0 && key < nk) {
        if (S.kmask && !S.kmask[(long)b * nk + key]) colacc = 0.f;   // masked keys had probability exactly 0
        S.out[((long)b * IMP_NUM_HEADS + h) * nk + key] = colacc;
    }
}

// Split-precision variant (hi/lo f16 halves, 3 f16 MFMAs per fp32 product - the arithmetic the attention kernels that
// produced `lse` use): K of the workgroup's 128 keys stays in registers as B-operand fragments, the 64-query tiles are
// staged through LDS as [hi | lo] rows (A operand, one ds_read_b128 per fragment).  5.3x less matrix-pipe time than the
// fp32 kernel above; same reduction order (in-lane over the accumulator registers, tiles in ascending order).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_colsum_f16x3_kernel(const ColsumParams p, int ktiles) {
    constexpr int KROW = DH + 4, KS = DH / 16, F4 = QT * DH / 4, LPT = F4 / 256;
    __shared__ __attribute__((aligned(16))) float Qs[2][QT][KROW];      // row = [DH hi halves | DH lo halves | pad]
    __shared__ __attribute__((aligned(16))) float Ls[2][QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int id = blockIdx.x;
    const int kt = id % ktiles; id /= ktiles;
    const int h = id % IMP_NUM_HEADS; id /= IMP_NUM_HEADS;
    const int sidx = id % p.nside;
    const int b = id / p.nside;
    const ColsumSide& S = p.side[sidx];
    const int nq = S.nq, nk = S.nk;
    const int k0 = kt * 128;
    if (k0 >= nk) return;
    const float* Qg = S.q + b * S.sq_b + h * DH;
    const float* Kg = S.k + b * S.sk_b + h * DH;
    const float* Lg = S.lse + ((long)b * IMP_NUM_HEADS + h) * (S.lq ? S.lq : nq);

    f16x8 kh[KS], kl[KS];                       // lane (key l31, half) holds d = 16 s + 8 half .. + 7 of k-step s
    {
        const int krow = k0 + wave * 32 + l31;
        const float* src = Kg + (long)(krow < nk ? krow : nk - 1) * p.ldk + 8 * half;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + 16 * s);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + 16 * s + 4);
            const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            u32x4 hh, ll;
#pragma unroll
            for (int i = 0; i < 4; ++i) { unsigned u0, u1; imp_split2(x[2 * i], x[2 * i + 1], u0, u1); hh[i] = u0; ll[i] = u1; }
            kh[s] = __builtin_bit_cast(f16x8, hh);
            kl[s] = __builtin_bit_cast(f16x8, ll);
        }
    }
    f32x4 rq[LPT];
    float rl = 0.f;
    auto load_tile = [&](int t) {
        const int q0 = t * QT;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int f = tid + j * 256;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            rq[j] = *reinterpret_cast<const f32x4*>(Qg + (long)min(q0 + row, nq - 1) * p.ldq + c4);   // lse = +inf masks the tail
        }
        if (tid < QT) rl = (q0 + tid < nq) ? Lg[q0 + tid] : INFINITY;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int f = tid + j * 256;
            const int row = f / (DH / 4), c4 = (f % (DH / 4)) * 4;
            u32x2 hi, lo;
#pragma unroll
            for (int i = 0; i < 2; ++i) { unsigned u0, u1; imp_split2(rq[j][2 * i], rq[j][2 * i + 1], u0, u1); hi[i] = u0; lo[i] = u1; }
            *reinterpret_cast<u32x2*>(&Qs[buf][row][c4 >> 1]) = hi;
            *reinterpret_cast<u32x2*>(&Qs[buf][row][DH / 2 + (c4 >> 1)]) = lo;
        }
        if (tid < QT) Ls[buf][tid] = rl;
    };
    float colacc = 0.f;
    const int nt = (nq + QT - 1) / QT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        f32x16 sacc[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[jb][r] = 0.f;
        f16x8 qh[2][KS], ql[2][KS];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float* qrow = &Qs[buf][jb * 32 + l31][4 * half + 8 * s];
                qh[jb][s] = *reinterpret_cast<const f16x8*>(qrow);
                ql[jb][s] = *reinterpret_cast<const f16x8*>(qrow + DH / 2);
            }
#pragma unroll
        for (int s = 0; s < KS; ++s) {          // rows (i) = queries, columns (j = lane) = keys
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[0][s], kh[s], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[1][s], kh[s], sacc[1], 0, 0, 0);
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[0][s], kl[s], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[1][s], kl[s], sacc[1], 0, 0, 0);
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[0][s], kh[s], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[1][s], kh[s], sacc[1], 0, 0, 0);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(&Ls[buf][jb * 32 + 8 * g + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = DH == 64 ? sacc[jb][4 * g + e] * 0.125f : sacc[jb][4 * g + e] / 5.656854249492381f;
                    colacc += __builtin_amdgcn_exp2f((s - ls[e]) * LOG2E);
                }
            }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }
    colacc += __shfl_xor(colacc, 32);
    const int key = k0 + wave * 32 + l31;
    if (half == 0 && key < nk) {
        if (S.kmask && !S.kmask[(long)b * nk + key]) colacc = 0.f;
        S.out[((long)b * IMP_NUM_HEADS + h) * nk + key] = colacc;
    }
}

// a[key] = (sum_h colsum[h][key]) / sum_key(...)   (nets/adgm.py:557-565), single workgroup, fixed order
__global__ __launch_bounds__(1024) void mass_normalize_kernel(const float* __restrict__ colsum, int n,
                                                              float* __restrict__ out) {
    __shared__ float part[16];
    __shared__ float total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float loc = 0.f;
    for (int i = tid; i < n; i += 1024) {
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < IMP_NUM_HEADS; ++h) a += colsum[(long)h * n + i];
        out[i] = a;
        loc += a;
    }
    loc = wave_sum(loc);
    if (lane == 0) part[wave] = loc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        total = t;
    }
    __syncthreads();
    const float t = total;
    for (int i = tid; i < n; i += 1024) out[i] = out[i] / t;
}

// ------------------------------------------------------------------------------------------------
// AdaGMN.pool selection, one 1024-thread workgroup per image side.
__device__ __forceinline__ unsigned f2key(float f) {      // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// k-th smallest (0-based) of vals[i] over i with flag[i] != 0 : 4-pass MSB radix select, LDS histogram
__device__ float radix_select(const float* __restrict__ vals, const unsigned char* flag, int n, int k,
                              unsigned* hist, unsigned* sh) {
    const int tid = threadIdx.x;
    unsigned prefix = 0, mask = 0;
    int kk = k;
    for (int pass = 3; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            if (flag[i]) {
                const unsigned key = f2key(vals[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {
            // first bin whose cumulative count exceeds kk, found by one wave: lane l owns bins 4l..4l+3, a shuffle scan
            // gives the counts before each lane (a single thread walking 256 LDS words cost ~10 us per pass)
            const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            unsigned incl = c0 + c1 + c2 + c3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (tid >= o) incl += up;
            }
            const unsigned long long over = __ballot(incl > (unsigned)kk);      // non-empty: the total exceeds kk
            const int owner = __ffsll((long long)over) - 1;
            if (tid == owner) {
                unsigned acc = incl - (c0 + c1 + c2 + c3);
                int bin = 4 * tid;
                if (acc + c0 > (unsigned)kk) { }
                else if (acc + c0 + c1 > (unsigned)kk) { acc += c0; bin += 1; }
                else if (acc + c0 + c1 + c2 > (unsigned)kk) { acc += c0 + c1; bin += 2; }
                else { acc += c0 + c1 + c2; bin += 3; }
                sh[0] = (unsigned)bin;
                sh[1] = acc;
            }
        }
        __syncthreads();
        prefix |= sh[0] << shift;
        mask |= 255u << shift;
        kk -= (int)sh[1];
        __syncthreads();
    }
    return key2f(prefix);
}

__global__ __launch_bounds__(1024) void pool_select_kernel(PoolSide s0, PoolSide s1, float thr, int32_t* counts) {
    extern __shared__ unsigned char dyn[];          // flag[n]
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[2];
    __shared__ int wcount[16];
    __shared__ int base_s, npid_s;
    const PoolSide& S = blockIdx.x == 0 ? s0 : s1;
    const int n = S.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* cnt = counts + 2 * blockIdx.x;
    if (S.skip) { if (tid == 0) { cnt[0] = -1; cnt[1] = 0; } return; }
    unsigned char* flag = dyn;
    if (tid == 0) npid_s = 0;
    __syncthreads();
    int loc = 0;
    for (int i = tid; i < n; i += 1024) {
        const unsigned char f = S.mass[i] >= thr ? 1 : 0;      // nets/adgm.py:578-579
        flag[i] = f;
        loc += f;
    }
    if (loc) atomicAdd(&npid_s, loc);
    __syncthreads();
    const int npid = npid_s;
    if (npid == 0) { if (tid == 0) { cnt[0] = -1; cnt[1] = 0; } return; }    // nets/adgm.py:580,586-587
    const int k = (npid - 1) / 2;                                           // torch.median = lower median
    const float md_self = radix_select(S.a_self, flag, n, k, hist, sh);
    const float md_cross = radix_select(S.a_cross, flag, n, k, hist, sh);
    // keep = pids | {a_self >= md_self} | {a_cross >= md_cross}; ascending ids (torch.unique sorts)
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + tid;
        const bool keep = i < n && (flag[i] || S.a_self[i] >= md_self || S.a_cross[i] >= md_cross);
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wcount[w];
        const int base = base_s;
        if (keep) S.ids[base + woff + before] = (int64_t)i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wcount[w];
            base_s = base + t;
        }
        __syncthreads();
    }
    if (tid == 0) { cnt[0] = base_s; cnt[1] = npid; }
}

__global__ __launch_bounds__(64) void gather_rows_kernel(const float* __restrict__ in, const int64_t* __restrict__ ids,
                                                         float* __restrict__ out, int n_in, int n_out, int dim) {
    const int b = blockIdx.y, i = blockIdx.x;
    const long src = ids[i];
    const f32x4* s = reinterpret_cast<const f32x4*>(in + ((long)b * n_in + src) * dim);
    f32x4* d = reinterpret_cast<f32x4*>(out + ((long)b * n_out + i) * dim);
    for (int c = threadIdx.x; c < dim / 4; c += 64) d[c] = s[c];
}

// masked AdaGMN, one pair of one iteration (nets/adgm.py:447-453 and :498-504): the matches found among the kept keypoints go back to
// full-size rows (out_i[g0[t]] = i0[t] >= 0 ? g1[i0[t]] : -1, out_m[g0[t]] = m0[t]); with ng0 / ng1 the kept id lists are composed with
// the pool's selection (ng[t] = keep ? g[keep[t]] : g[t]) and entered into the key masks of the next layers.  Id lists hold unique ids.
__global__ __launch_bounds__(256) void masked_commit_kernel(const MaskedCommit p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < p.n0sel) {
        const long g = p.g0[t], i = p.i0[t];
        p.out_i[g] = i >= 0 ? p.g1[i] : -1;
        p.out_m[g] = p.m0[t];
    }
    if (p.ng0 && t < p.nk0) {
        const long g = p.keep0 ? p.g0[p.keep0[t]] : p.g0[t];
        p.ng0[t] = g;
        p.mask0[g] = 1;
    }
    if (p.ng1 && t < p.nk1) {
        const long g = p.keep1 ? p.g1[p.keep1[t]] : p.g1[t];
        p.ng1[t] = g;
        p.mask1[g] = 1;
    }
}

}  // namespace

hipError_t launch_normalize_kpts(const float* kpts, long count, float width, float height, float* out,
                                 hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const float mx = width > height ? width : height;
    const float scaling = mx * 0.7f;                 // size.max() * 0.7 in fp32 (nets/layers.py:55)
    hipLaunchKernelGGL(normalize_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, kpts, count, width / 2.f,
                       height / 2.f, scaling, out);
    return hipGetLastError();
}

hipError_t launch_kenc_first(const Kenc0Side sides[2], int batch, int c0, const float* W0, const float* b0,
                             float width, float height, hipStream_t stream, const RaggedCounts* rc) {
    const int nmax = sides[0].n > sides[1].n ? sides[0].n : sides[1].n;
    if (nmax <= 0 || c0 > 64) return c0 > 64 ? hipErrorInvalidValue : hipSuccess;
    float scaling = 0.f;
    if (width > 0.f) scaling = (width > height ? width : height) * 0.7f;
    RaggedCounts r;
    if (rc) r = *rc; else r.on = 0;
    hipLaunchKernelGGL(kenc_first_kernel, dim3((nmax + 127) / 128, 2, batch), dim3(128), 0, stream, sides[0], sides[1], c0,
                       W0, b0, width / 2.f, height / 2.f, scaling, r);
    return hipGetLastError();
}

hipError_t launch_attn_colsum(const ColsumParams& p, int batch, int prec, hipStream_t stream) {
    int maxk = p.side[0].nk;
    if (p.nside == 2 && p.side[1].nk > maxk) maxk = p.side[1].nk;
    if (maxk <= 0) return hipSuccess;
    const int ktiles = (maxk + 127) / 128;
    const int total = ktiles * IMP_NUM_HEADS * p.nside * batch;
    if (prec == 1) {        // split-half f16x3 products, like the attention that produced lse
        if (p.dh == 64) hipLaunchKernelGGL(attn_colsum_f16x3_kernel<64>, dim3(total), dim3(256), 0, stream, p, ktiles);
        else if (p.dh == 32) hipLaunchKernelGGL(attn_colsum_f16x3_kernel<32>, dim3(total), dim3(256), 0, stream, p, ktiles);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.dh == 64) hipLaunchKernelGGL(attn_colsum_kernel<64>, dim3(total), dim3(256), 0, stream, p, ktiles);
    else if (p.dh == 32) hipLaunchKernelGGL(attn_colsum_kernel<32>, dim3(total), dim3(256), 0, stream, p, ktiles);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_attn_mass_normalize(const float* colsum, int n, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(mass_normalize_kernel, dim3(1), dim3(1024), 0, stream, colsum, n, out);
    return hipGetLastError();
}

hipError_t launch_pool_select(const PoolSide sides[2], int nsides, float thr, int32_t* counts, hipStream_t stream) {
    const int nmax = (nsides == 2 && sides[1].n > sides[0].n) ? sides[1].n : sides[0].n;
    hipLaunchKernelGGL(pool_select_kernel, dim3(nsides), dim3(1024), (size_t)((nmax + 15) & ~15), stream, sides[0],
                       sides[nsides - 1], thr, counts);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* in, const int64_t* ids, float* out, int batch, int n_in, int n_out, int dim,
                              hipStream_t stream) {
    if (n_out <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_out, batch), dim3(64), 0, stream, in, ids, out, n_in, n_out, dim);
    return hipGetLastError();
}

hipError_t launch_masked_commit(const MaskedCommit& p, hipStream_t stream) {
    int n = p.n0sel;
    if (p.ng0 && p.nk0 > n) n = p.nk0;
    if (p.ng1 && p.nk1 > n) n = p.nk1;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(masked_commit_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}
