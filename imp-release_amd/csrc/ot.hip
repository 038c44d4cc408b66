// Optimal-transport scoring and match extraction for gfx950.
//   sink_algorithm / sinkhorn   nets/layers.py:27-46   (probability domain, NOT log domain)
//   dual_softmax                nets/layers.py:20-24
//   GM.compute_matches          nets/gm.py:305-320
//
// HBM/LLC-bound byte work: the (N+1)x(M+1) fp32 matrix (16.8 MB at N=M=2048, 67 MB for 4 pairs: Infinity-Cache
// resident) is the only large operand of the T Sinkhorn iterations.
//
// Main path (rows up to 3328 floats): ONE read of P per iteration.  ot_fused_pass_kernel holds a row in registers,
// computes u_i = r_i / (P_i . v + eps) and immediately accumulates P_ij * u_i into per-lane column partials, so the
// transposed mat-vec needs no second pass; ot_colreduce_kernel sums the per-workgroup partial vectors (plus the
// constant dustbin row) into v.  Two launches per iteration over all B pairs: a kernel boundary (~1.5 us) is the
// cheapest chip-wide dependency on this part (a grid barrier costs 4-6 us).
// Fallback (longer rows): P and its transpose P^T are both kept and every half-iteration is a coalesced float4 row
// pass with a wavefront shuffle reduction (ot_rowpass_kernel).
// No atomics and fixed summation orders everywhere: results are bit-reproducible run to run.
#include "imp_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 12-byte record of the 3-byte matrix copy (a 3-element ext_vector would be padded to 16 bytes)
struct __attribute__((packed, aligned(4))) u32x3 { unsigned w0, w1, w2; };

namespace {

// ---- 3-byte storage of P for the Sinkhorn iterations ---------------------------------------------------------
// The iterations only need the scaling vectors u, v; they stream the matrix T times, so its bytes are the cost.  A copy
// that keeps sign, exponent and 15 mantissa bits (round to nearest even: relative 2^-17) moves 3/4 of the bytes; the
// scores p.u.v and the maxima are still formed from the fp32 matrix.  Chosen by experiment like the f16x3 products:
// emulated in the oracle on 11 pairs (N = 512 / 1024 / 2048, L = 9, T = 100) it gives 0 index mismatches and
// |dmscore| <= 1.0e-5 (tolerance 1e-4), whereas 11 mantissa bits (a 2-byte copy) drifts to 1.8e-4.
// Four consecutive values a, b, c, d are packed into three dwords: [a0 a1 a2 b0 | b1 b2 c0 c1 | c2 d0 d1 d2] (x2 = top byte).
__device__ __forceinline__ unsigned q24(float x) {          // rounded fp32 bit pattern with the low byte cleared
    unsigned u = __float_as_uint(x);
    u += 0x7Fu + ((u >> 8) & 1u);
    return u & 0xFFFFFF00u;
}
__device__ __forceinline__ u32x3 pack24(const f32x4 x) {
    const unsigned a = q24(x[0]) >> 8, b = q24(x[1]) >> 8, c = q24(x[2]) >> 8, d = q24(x[3]) >> 8;
    return u32x3{a | (b << 24), (b >> 8) | (c << 16), (c >> 16) | (d << 8)};   // aggregate init of the 12-byte record
}
__device__ __forceinline__ f32x4 unpack24(const u32x3 w) {
    return f32x4{__uint_as_float(w.w0 << 8),
                 __uint_as_float(__builtin_amdgcn_perm(w.w1, w.w0, 0x0504030cu)),     // v_perm_b32: 0-3 = 2nd source, 0x0c = 0
                 __uint_as_float(__builtin_amdgcn_perm(w.w2, w.w1, 0x0403020cu)),
                 __uint_as_float(w.w2 & 0xFFFFFF00u)};
}

constexpr float OT_EPS = 1e-8f;   // nets/layers.py:13

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// (value, index) max with "first index wins" on ties (torch.max semantics, nets/gm.py:306-307)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// ---- init: dustbin augmentation (nets/layers.py:39-40) + row softmax (nets/layers.py:28) -------------
// one wave per augmented row; dual != 0 keeps the augmented logits instead (dual_softmax needs them)
template <int NV>     // NV > 0: ceil(ldp / 64) columns per lane held in registers; 0: streaming form (any row length)
__global__ __launch_bounds__(256) void ot_init_kernel(const float* __restrict__ dist, int n0, int n1, float bin,
                                                      int dual, float* __restrict__ P, int ldp,
                                                      float* __restrict__ u, float* __restrict__ v, int ldpt,
                                                      float* __restrict__ v2, unsigned* __restrict__ P24) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const int b = blockIdx.y;
    if (row > n0) return;
    const float* drow = dist + ((long)b * n0 + row) * n1;
    float* prow = P + ((long)b * (n0 + 1) + row) * ldp;
    const bool last = row == n0;
    if (row == 0) {   // Sinkhorn start vectors (nets/layers.py:29-30); padded tails are zero
        for (int j = lane; j < ldp; j += 64) {
            v[(long)b * ldp + j] = j <= n1 ? 1.f : 0.f;
            if (v2) v2[(long)b * ldp + j] = 0.f;      // pads of the ping-pong buffer stay zero
        }
        for (int i = lane; i < ldpt; i += 64) u[(long)b * ldpt + i] = i <= n0 ? 1.f : 0.f;
    }
    if (dual) {
        for (int j = lane; j < ldp; j += 64) prow[j] = j > n1 ? 0.f : ((last || j == n1) ? bin : drow[j]);
        return;
    }
    if (NV > 0 && P24 == nullptr) {
        // the row lives in registers (lane owns columns lane, lane + 64, ...): one read of dist, one exp per element;
        // same values and the same summation order as the streaming form below
        float x[NV > 0 ? NV : 1];
        float mx = bin;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = lane + 64 * k;
            x[k] = j > n1 ? -INFINITY : ((last || j == n1) ? bin : drow[j]);
            if (j < n1 && !last) mx = fmaxf(mx, x[k]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            x[k] = expf(x[k] - mx);             // exp(-inf) = 0 for the padded tail
            if (lane + 64 * k <= n1) sum += x[k];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = lane + 64 * k;
            if (j < ldp) prow[j] = j > n1 ? 0.f : x[k] / sum;
        }
        return;
    }
    float mx = bin;
    if (!last) for (int j = lane; j < n1; j += 64) mx = fmaxf(mx, drow[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= n1; j += 64) sum += expf(((last || j == n1) ? bin : drow[j]) - mx);
    sum = wave_sum(sum);
    if (P24 == nullptr) {
        for (int j = lane; j < ldp; j += 64)
            prow[j] = j > n1 ? 0.f : expf(((last || j == n1) ? bin : drow[j]) - mx) / sum;
        return;
    }
    // 4 consecutive columns per lane: one float4 of P and the 12-byte packed copy (same values, same expressions)
    u32x3* qrow = reinterpret_cast<u32x3*>(P24 + ((long)b * (n0 + 1) + row) * (3 * (ldp >> 2)));
    for (int j4 = lane; j4 < (ldp >> 2); j4 += 64) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * j4 + e;
            o[e] = j > n1 ? 0.f : expf(((last || j == n1) ? bin : drow[min(j, n1 - 1)]) - mx) / sum;
        }
        *reinterpret_cast<f32x4*>(prow + 4 * j4) = o;
        qrow[j4] = pack24(o);
    }
}

// ---- 32x32 LDS tile transpose: PT[j][i] = P[i][j]; padded tail of PT rows zeroed ----------------------
__global__ __launch_bounds__(256) void ot_transpose_kernel(const float* __restrict__ P, int rows, int cols, int ldp,
                                                           float* __restrict__ PT, int ldpt) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float* src = P + (long)b * rows * ldp;
    float* dst = PT + (long)b * cols * ldpt;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[(long)r * ldp + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;   // dst row = c (source column), dst col = r
        if (c < cols && r < ldpt) dst[(long)c * ldpt + r] = tile[tx][ty + 8 * k];
    }
}

// ---- one Sinkhorn half-iteration (nets/layers.py:32 or :33) as a row pass --------------------------------
//   out[i] = marg_i / (sum_j Mx[i][j] * in[j] + eps),  marg = 1 except the last (dustbin) row = rows
// RPW rows per wave share the loaded `in` vector chunk.
template <int RPW>
__global__ __launch_bounds__(256) void ot_rowpass_kernel(const float* __restrict__ Mx, int rows, int ld,
                                                         const float* __restrict__ in, float* __restrict__ out,
                                                         int ld_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int r0 = (blockIdx.x * 4 + wave) * RPW;
    if (r0 >= rows) return;
    const float* base = Mx + (long)b * rows * ld;
    const f32x4* vin = reinterpret_cast<const f32x4*>(in + (long)b * ld);
    const f32x4* rp[RPW];
    float acc[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r0 + k < rows ? r0 + k : rows - 1;
        rp[k] = reinterpret_cast<const f32x4*>(base + (long)r * ld);
        acc[k] = 0.f;
    }
    const int n4 = ld >> 2;
    for (int c = lane; c < n4; c += 64) {
        const f32x4 x = vin[c];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const f32x4 m = rp[k][c];
            acc[k] = fmaf(m[0], x[0], acc[k]);
            acc[k] = fmaf(m[1], x[1], acc[k]);
            acc[k] = fmaf(m[2], x[2], acc[k]);
            acc[k] = fmaf(m[3], x[3], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const float s = wave_sum(acc[k]);
        const int r = r0 + k;
        if (lane == 0 && r < rows) {
            const float marg = r == rows - 1 ? (float)rows : 1.f;
            out[(long)b * ld_out + r] = marg / (s + OT_EPS);
        }
    }
}

// ---- fused Sinkhorn iteration: ONE read of P per iteration ------------------------------------------------
// A wave keeps a whole row of P in registers (NCH float4 per lane): dot with v -> u_i (nets/layers.py:32), then the
// SAME registers times u_i are accumulated into per-lane column partials for nets/layers.py:33, so the transposed
// pass disappears.  4 rows per wave (next row prefetched while the current one is reduced), 4 waves per workgroup
// combine their partials in LDS and write one partial vector per workgroup; ot_colreduce_kernel turns the partial
// vectors into v.  Fixed summation order everywhere (no atomics).
constexpr int FP_WAVES = 8, FP_RPW = 4, FP_ROWS = FP_WAVES * FP_RPW;   // 512 threads, 32 rows per workgroup
constexpr int FP_MAX_LD = 3328;   // 13 float4 per lane: 2 row slots + accumulators fill the 256-VGPR budget (17 spills), (1 + 8) x ld floats = 117 KB of LDS
template <int NCH, int COMPACT>
__global__ __launch_bounds__(512, 2) void ot_fused_pass_kernel(const float* __restrict__ P, int rows, int prows, int ld,
                                                               const float* __restrict__ v, float* __restrict__ u,
                                                               int ld_u, float* __restrict__ partials, int nwg) {
    // rows = n0 REAL rows handled here (the constant dustbin row n0 is folded into ot_colreduce_kernel so that the
    // grid is ceil(n0/16) x B workgroups - exactly 2 per CU at n0 = 2048, B = 4); prows = n0 + 1 rows per pair in P
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [ld] v  |  [FP_WAVES][ld] partial vectors
    float* vs = lds;
    float* red = lds + ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * FP_ROWS + wave * FP_RPW;
    const float* base = P + (long)b * prows * ld;
    const int n4 = ld >> 2;
    constexpr int NSLOT = 2;
    f32x4 part[NCH], row[NSLOT][NCH];
    auto load_row = [&](int slot, int r) {
        if (COMPACT) {      // P points at the 3-byte copy: 3 dwords per 4 values
            const u32x3* rp = reinterpret_cast<const u32x3*>(reinterpret_cast<const unsigned*>(P) +
                                                             ((long)b * prows + min(r, rows - 1)) * (3 * n4));
#pragma unroll
            for (int c = 0; c < NCH; ++c) row[slot][c] = unpack24(rp[min(lane + 64 * c, n4 - 1)]);
            return;
        }
        const f32x4* rp = reinterpret_cast<const f32x4*>(base + (long)min(r, rows - 1) * ld);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int c4 = min(lane + 64 * c, n4 - 1);     // clamped: the duplicate is zeroed by the v image below
            row[slot][c] = rp[c4];
        }
    };
    load_row(0, r0);
    if (NSLOT == 2) load_row(1, r0 + 1);
    {   // v -> LDS once per workgroup; chunk slots past the row end read an explicit zero
        const f32x4* vin = reinterpret_cast<const f32x4*>(v + (long)b * ld);
        for (int c4 = threadIdx.x; c4 < n4; c4 += 512) *reinterpret_cast<f32x4*>(vs + 4 * c4) = vin[c4];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) part[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FP_RPW; ++k) {
        const int r = r0 + k;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int c4 = lane + 64 * c;
            if (c4 < n4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(vs + 4 * c4);
                const f32x4 m = row[k % NSLOT][c];
                acc = fmaf(m[0], x[0], acc);
                acc = fmaf(m[1], x[1], acc);
                acc = fmaf(m[2], x[2], acc);
                acc = fmaf(m[3], x[3], acc);
            }
        }
        const float sdot = wave_sum(acc);
        const float ui = r < rows ? 1.f / (sdot + OT_EPS) : 0.f;      // real rows: marginal 1; rows past the end: nothing
        if (lane == 0 && r < rows) u[(long)b * ld_u + r] = ui;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[c][e] = fmaf(row[k % NSLOT][c][e], ui, part[c][e]);
        if (k + NSLOT < FP_RPW) load_row(k % NSLOT, r0 + k + NSLOT);      // rotate the register slots: the next row streams in
    }
    // workgroup combine: FP_WAVES partial vectors -> 1 (fixed order)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int c4 = lane + 64 * c;
        if (c4 < n4) *reinterpret_cast<f32x4*>(red + wave * ld + 4 * c4) = part[c];
    }
    __syncthreads();
    float* out = partials + ((long)b * nwg + blockIdx.x) * ld;
    for (int j = threadIdx.x; j < ld; j += 512) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FP_WAVES; ++w) s += red[w * ld + j];
        out[j] = s;
    }
}

// v_new[j] = c_j / (sum over workgroup partials + P[n0][j] * u[n0] + eps), c = 1 except the dustbin column = cols
// (nets/layers.py:33,43-44).  The dustbin ROW n0 of P is handled here: u[n0] = (n0+1) / (P[n0][:] . v_old + eps)
// (nets/layers.py:32,41-42), computed redundantly per workgroup in a fixed order; v is ping-ponged (v_old -> v_new).
// block = 64 columns x 16 partial groups (1024 threads): group g sums partials g, g+16, ... ; LDS combine in order
template <int COMPACT>
__global__ __launch_bounds__(1024) void ot_colreduce_kernel(const float* __restrict__ partials, int nwg, int ld, int cols,
                                                            const float* __restrict__ P, int n0,
                                                            const float* __restrict__ v_old, float* __restrict__ v_new,
                                                            float* __restrict__ u, int ld_u) {
    __shared__ float sm[16][64];
    __shared__ float dpart[16];
    __shared__ float ulast_s;
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int j = blockIdx.x * 64 + cx;
    const float* plast = P + ((long)b * (n0 + 1) + n0) * ld;
    const float* vo = v_old + (long)b * ld;
    // dustbin-row dot product: thread t takes columns t, t+1024, ... (pads of P and v are zero)
    float d = 0.f;
    // (compact: the iterations see the 3-byte values of the whole matrix, so the dustbin row is rounded the same way)
    for (int c = threadIdx.x; c < ld; c += 1024) d = fmaf(COMPACT ? __uint_as_float(q24(plast[c])) : plast[c], vo[c], d);
    float s = 0.f;
    if (j < cols) {
        const float* pp = partials + (long)b * nwg * ld + j;
        int w = g;
        for (; w + 48 < nwg; w += 64) {          // 4 independent loads in flight
            const float t0 = pp[(long)w * ld], t1 = pp[(long)(w + 16) * ld], t2 = pp[(long)(w + 32) * ld],
                        t3 = pp[(long)(w + 48) * ld];
            s += (t0 + t1) + (t2 + t3);
        }
        for (; w < nwg; w += 16) s += pp[(long)w * ld];
    }
    d = wave_sum(d);
    sm[g][cx] = s;
    if (lane == 0) dpart[g] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += dpart[k];
        const float ul = (float)(n0 + 1) / (t + OT_EPS);
        ulast_s = ul;
        if (blockIdx.x == 0) u[(long)b * ld_u + n0] = ul;
    }
    __syncthreads();
    if (g == 0 && j < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cx];
        t = fmaf(COMPACT ? __uint_as_float(q24(plast[j])) : plast[j], ulast_s, t);
        const float marg = j == cols - 1 ? (float)cols : 1.f;
        v_new[(long)b * ld + j] = marg / (t + OT_EPS);
    }
}

// ---- dual softmax: per-row log-sum-exp of the logits (rows of P -> u, rows of PT -> v) -----------------
__global__ __launch_bounds__(256) void ot_rowlse_kernel(const float* __restrict__ Mx, int rows, int cols, int ld,
                                                        float* __restrict__ out, int ld_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* row = Mx + ((long)b * rows + r) * ld;
    float mx = -INFINITY;
    for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, row[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) s += expf(row[j] - mx);
    s = wave_sum(s);
    if (lane == 0) out[(long)b * ld_out + r] = mx + logf(s);
}

__device__ __forceinline__ float ot_value(float m, float ui, float vj, int dual) {
    // Sinkhorn: (p * u) * v as nets/layers.py:34 ; dual: exp(log_softmax_row + log_softmax_col) nets/layers.py:23-24
    return dual ? expf((m - ui) + (m - vj)) : (m * ui) * vj;
}

__global__ __launch_bounds__(256) void ot_scores_kernel(const float* __restrict__ P, int n0, int n1, int ldp,
                                                        const float* __restrict__ u, int ldu,
                                                        const float* __restrict__ v, int dual,
                                                        float* __restrict__ scores) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + wave;
    if (i > n0) return;
    const float* prow = P + ((long)b * (n0 + 1) + i) * ldp;
    const float ui = u[(long)b * ldu + i];
    const float* vb = v + (long)b * ldp;
    float* srow = scores + ((long)b * (n0 + 1) + i) * (n1 + 1);
    for (int j = lane; j <= n1; j += 64) srow[j] = ot_value(prow[j], ui, vb[j], dual);
}

// row maxima of the inner block straight from P (TRANSPOSED = 0) or PT (TRANSPOSED = 1); both evaluate
// the identical expression (p*u_i)*v_j so the two sides see bit-identical values for the mutual check.
template <int TRANSPOSED>
__global__ __launch_bounds__(256) void ot_maxima_kernel(const float* __restrict__ Mx, int rows, int cols, int ld,
                                                        const float* __restrict__ rowvec, int ldr,
                                                        const float* __restrict__ colvec, int ldc, int dual,
                                                        float* __restrict__ maxv, int* __restrict__ argv) {
    // rows/cols = inner sizes (n0,n1) or (n1,n0); Mx has rows+1 rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* row = Mx + ((long)b * (rows + 1) + r) * ld;
    const float rvv = rowvec[(long)b * ldr + r];
    const float* cv = colvec + (long)b * ldc;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float val = TRANSPOSED ? ot_value(row[c], cv[c], rvv, dual) : ot_value(row[c], rvv, cv[c], dual);
        if (val > best) { best = val; bi = c; }
    }
    wave_argmax(best, bi);
    if (lane == 0) { maxv[(long)b * rows + r] = best; argv[(long)b * rows + r] = bi; }
}

// ---- maxima of an arbitrary score tensor [B][n0+1][n1+1] --------------------------------------------------
__global__ __launch_bounds__(256) void score_rowmax_kernel(const float* __restrict__ scores, int n0, int n1,
                                                           float* __restrict__ maxv, int* __restrict__ argv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n0) return;
    const float* row = scores + ((long)b * (n0 + 1) + r) * (n1 + 1);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    bool nan = false;                     // a NaN score voids the row's maximum (torch.max propagates NaN; a voided resident Sinkhorn
    for (int c = lane; c < n1; c += 64) { // launch poisons its rows with NaN: the matches must come out void, not plausible)
        const float val = row[c];
        nan |= val != val;
        if (val > best) { best = val; bi = c; }
    }
    wave_argmax(best, bi);
    if (__any(nan)) { best = __builtin_nanf(""); bi = 0x7fffffff; }
    if (lane == 0) { maxv[(long)b * n0 + r] = best; argv[(long)b * n0 + r] = bi; }
}

constexpr int COL_CHUNK = 64;   // rows per column-maximum / column-sum partial
// partial column maxima over a chunk of 64 rows: block = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void score_colmax_part_kernel(const float* __restrict__ scores, int n0, int n1,
                                                                int chunks, float* __restrict__ pv,
                                                                int* __restrict__ pi) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.z, chunk = blockIdx.y;
    const int c = blockIdx.x * 64 + cx;
    const float* base = scores + (long)b * (n0 + 1) * (n1 + 1);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (c < n1) {
        const int rend = min(n0, (chunk + 1) * COL_CHUNK);
        for (int r = chunk * COL_CHUNK + rg; r < rend; r += 4) {
            const float val = base[(long)r * (n1 + 1) + c];
            if (val != val) { best = val; bi = 0x7fffffff; break; }      // NaN voids the column (sticky: see score_rowmax_kernel)
            if (val > best) { best = val; bi = r; }
        }
    }
    sv[rg][cx] = best; si[rg][cx] = bi;
    __syncthreads();
    if (rg == 0 && c < n1) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float ov = sv[k][cx]; const int oi = si[k][cx];
            if (best != best) break;
            if (ov != ov) { best = ov; bi = 0x7fffffff; break; }
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        pv[((long)b * chunks + chunk) * n1 + c] = best;
        pi[((long)b * chunks + chunk) * n1 + c] = bi;
    }
}
__global__ __launch_bounds__(256) void colmax_combine_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                             int n1, int chunks, float* __restrict__ maxv,
                                                             int* __restrict__ argv) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n1) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = 0; k < chunks; ++k) {   // ascending row chunks: strict > keeps the first maximal row
        const float ov = pv[((long)b * chunks + k) * n1 + c];
        const int oi = pi[((long)b * chunks + k) * n1 + c];
        if (ov != ov) { best = ov; bi = 0x7fffffff; break; }             // a NaN partial voids the column
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    maxv[(long)b * n1 + c] = best; argv[(long)b * n1 + c] = bi;
}

// ---- mutual nearest neighbours + threshold (nets/gm.py:308-318) ---------------------------------------------
__global__ __launch_bounds__(256) void mutual_kernel(int ld0, int ld1, const float* __restrict__ max0,
                                                     const int* __restrict__ arg0, const float* __restrict__ max1,
                                                     const int* __restrict__ arg1, float p,
                                                     int64_t* __restrict__ ind0, int64_t* __restrict__ ind1,
                                                     float* __restrict__ ms0, float* __restrict__ ms1, int* __restrict__ range_flag, RaggedCounts rc) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    // ragged batches: pair b has n0 x n1 keypoints inside arrays padded to ld0 / ld1; keypoints past its own count are "unmatched"
    const int n0 = imp_count(rc, 0, b, ld0), n1 = imp_count(rc, 1, b, ld1);
    if (rc.on) {
        if (t >= n0 && t < ld0) { if (ms0) ms0[(long)b * ld0 + t] = 0.f; if (ind0) ind0[(long)b * ld0 + t] = -1; }
        if (t >= n1 && t < ld1) { if (ms1) ms1[(long)b * ld1 + t] = 0.f; if (ind1) ind1[(long)b * ld1 + t] = -1; }
    }
    const float* m0 = max0 + (long)b * ld0;
    const int* a0 = arg0 + (long)b * ld0;
    const int* a1 = arg1 + (long)b * ld1;
    // an argmax of 0x7fffffff means "no maximum found" (a row of NaNs: |operand| >= 65504 in f16x3 mode, or NaN inputs):
    // such a keypoint has no match - never an out-of-range read
    // a row / column maximum of -inf: every score of that row was NaN (an MFMA operand beyond the fp16 range in f16x3 mode, or
    // non-finite inputs).  The matches come out as -1; the flag makes the library say so at its next entry (IMP_E_RANGE)
    // (a NaN maximum: the maxima were taken from a score TENSOR that holds NaN - the same causes, or rows poisoned by a voided resident
    // Sinkhorn launch; the library looks at the resident word first, so a void is reported as IMP_E_RESIDENT, not as a range error)
    if (range_flag && t < n0 && (m0[t] == -INFINITY || m0[t] != m0[t])) __hip_atomic_store(range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t < n0) {
        const int j = a0[t];
        const bool mutual = (unsigned)j < (unsigned)n1 && a1[j] == t;
        float s = mutual ? m0[t] : 0.f;
        if (j == 0x7fffffff && m0[t] != m0[t]) s = m0[t];       // voided by the resident Sinkhorn kernel (time-out): NaN, never a plausible 0
        if (ms0) ms0[(long)b * ld0 + t] = s;
        if (ind0) ind0[(long)b * ld0 + t] = (mutual && s > p) ? (int64_t)j : (int64_t)-1;
    }
    if (t < n1) {
        const int i = a1[t];
        const bool ok_i = (unsigned)i < (unsigned)n0;
        const int ji = ok_i ? a0[i] : -1;
        const bool mutual1 = ok_i && ji == t;
        const bool mutual0_i = ok_i && (unsigned)ji < (unsigned)n1 && a1[ji] == i;
        const float s0_i = mutual0_i ? m0[i] : 0.f;
        const bool valid0_i = mutual0_i && s0_i > p;
        const bool voided = i == 0x7fffffff && max1[(long)b * ld1 + t] != max1[(long)b * ld1 + t];
        if (ms1) ms1[(long)b * ld1 + t] = voided ? max1[(long)b * ld1 + t] : (mutual1 ? s0_i : 0.f);
        if (ind1) ind1[(long)b * ld1 + t] = (mutual1 && valid0_i) ? (int64_t)i : (int64_t)-1;
    }
}

// ---- inner-block row sums / column sums of a score tensor (pooling confidence, nets/adgm.py:577-579,592-593)
__global__ __launch_bounds__(256) void score_rowsum_kernel(const float* __restrict__ scores, int n0, int n1,
                                                           float* __restrict__ mass0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n0) return;
    const float* row = scores + (long)r * (n1 + 1);
    float s = 0.f;
    for (int c = lane; c < n1; c += 64) s += row[c];
    s = wave_sum(s);
    if (lane == 0) mass0[r] = s;
}
__global__ __launch_bounds__(256) void score_colsum_part_kernel(const float* __restrict__ scores, int n0, int n1,
                                                                float* __restrict__ part) {
    __shared__ float sv[4][64];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int chunk = blockIdx.y;
    const int c = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (c < n1) {
        const int rend = min(n0, (chunk + 1) * COL_CHUNK);
        for (int r = chunk * COL_CHUNK + rg; r < rend; r += 4) s += scores[(long)r * (n1 + 1) + c];
    }
    sv[rg][cx] = s;
    __syncthreads();
    if (rg == 0 && c < n1) part[(long)chunk * n1 + c] = (sv[0][cx] + sv[1][cx]) + (sv[2][cx] + sv[3][cx]);
}
__global__ __launch_bounds__(256) void colsum_combine_kernel(const float* __restrict__ part, int n1, int chunks,
                                                             float* __restrict__ mass1) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n1) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(long)k * n1 + c];
    mass1[c] = s;
}

}  // namespace

hipError_t launch_ot_init(const float* dist, int batch, int n0, int n1, float bin_score, int dual,
                          const OtBuffers& ot, hipStream_t stream) {
    const bool compact = ot.compact && ot.P24 && !dual && ot.partials && ot.v2 && ot.ldp <= FP_MAX_LD;   // fused-path launches only
    unsigned* p24 = compact ? ot.P24 : (unsigned*)nullptr;
    const dim3 grid((n0 + 1 + 3) / 4, batch);
#define IMP_OT_INIT(NV) hipLaunchKernelGGL(ot_init_kernel<NV>, grid, dim3(256), 0, stream, dist, n0, n1, bin_score, dual, \
                                           ot.P, ot.ldp, ot.u, ot.v, ot.ldpt, ot.v2, p24)
    if (dual || compact || ot.ldp > 3328) IMP_OT_INIT(0);
    else if (ot.ldp <= 512) IMP_OT_INIT(8);
    else if (ot.ldp <= 1280) IMP_OT_INIT(20);
    else if (ot.ldp <= 2304) IMP_OT_INIT(36);
    else IMP_OT_INIT(52);
#undef IMP_OT_INIT
    hipLaunchKernelGGL(ot_transpose_kernel, dim3((n1 + 1 + 31) / 32, (ot.ldpt + 31) / 32, batch), dim3(256), 0, stream,
                       ot.P, n0 + 1, n1 + 1, ot.ldp, ot.PT, ot.ldpt);
    return hipGetLastError();
}

template <int NCH, int COMPACT>
static void launch_fused_iteration(int batch, int n0, int n1, OtBuffers& ot, hipStream_t stream) {
    const int nwg = (n0 + FP_ROWS - 1) / FP_ROWS;
    const size_t lds = (size_t)(1 + FP_WAVES) * ot.ldp * sizeof(float);
    (void)imp_grant_dynamic_lds((const void*)ot_fused_pass_kernel<NCH, COMPACT>, (size_t)(1 + FP_WAVES) * (NCH * 256) * sizeof(float));
    hipLaunchKernelGGL((ot_fused_pass_kernel<NCH, COMPACT>), dim3(nwg, batch), dim3(512), lds, stream,
                       COMPACT ? reinterpret_cast<const float*>(ot.P24) : ot.P, n0, n0 + 1, ot.ldp, ot.v, ot.u, ot.ldpt,
                       ot.partials, nwg);
    hipLaunchKernelGGL(ot_colreduce_kernel<COMPACT>, dim3((n1 + 1 + 63) / 64, batch), dim3(1024), 0, stream, ot.partials, nwg,
                       ot.ldp, n1 + 1, ot.P, n0, ot.v, ot.v2, ot.u, ot.ldpt);
    float* t = ot.v; ot.v = ot.v2; ot.v2 = t;
}

hipError_t launch_ot_iterations(int batch, int n0, int n1, int iterations, OtBuffers& ot, hipStream_t stream) {
    if (ot.partials && ot.v2 && ot.ldp <= FP_MAX_LD) {
        // fused path: P is read once per iteration (row held in registers), column partials reduced by a tiny kernel
        const bool compact = ot.compact && ot.P24;
        for (int it = 0; it < iterations; ++it) {
            if (compact) {
                if (ot.ldp <= 512) launch_fused_iteration<2, 1>(batch, n0, n1, ot, stream);
                else if (ot.ldp <= 1280) launch_fused_iteration<5, 1>(batch, n0, n1, ot, stream);
                else if (ot.ldp <= 2304) launch_fused_iteration<9, 1>(batch, n0, n1, ot, stream);
                else launch_fused_iteration<13, 1>(batch, n0, n1, ot, stream);
            } else {
                if (ot.ldp <= 512) launch_fused_iteration<2, 0>(batch, n0, n1, ot, stream);
                else if (ot.ldp <= 1280) launch_fused_iteration<5, 0>(batch, n0, n1, ot, stream);
                else if (ot.ldp <= 2304) launch_fused_iteration<9, 0>(batch, n0, n1, ot, stream);
                else launch_fused_iteration<13, 0>(batch, n0, n1, ot, stream);
            }
        }
        return hipGetLastError();
    }
    constexpr int RPW = 2;
    const dim3 g0((n0 + 1 + 4 * RPW - 1) / (4 * RPW), batch), g1((n1 + 1 + 4 * RPW - 1) / (4 * RPW), batch);
    for (int it = 0; it < iterations; ++it) {
        // u = r / (P v + eps)        nets/layers.py:32   (u lives with ld = ldpt: it multiplies PT's columns)
        hipLaunchKernelGGL(ot_rowpass_kernel<RPW>, g0, dim3(256), 0, stream, ot.P, n0 + 1, ot.ldp, ot.v, ot.u, ot.ldpt);
        // v = c / (P^T u + eps)      nets/layers.py:33
        hipLaunchKernelGGL(ot_rowpass_kernel<RPW>, g1, dim3(256), 0, stream, ot.PT, n1 + 1, ot.ldpt, ot.u, ot.v, ot.ldp);
    }
    return hipGetLastError();
}

hipError_t launch_ot_dual_lse(int batch, int n0, int n1, const OtBuffers& ot, hipStream_t stream) {
    hipLaunchKernelGGL(ot_rowlse_kernel, dim3((n0 + 1 + 3) / 4, batch), dim3(256), 0, stream, ot.P, n0 + 1, n1 + 1, ot.ldp,
                       ot.u, ot.ldpt);
    hipLaunchKernelGGL(ot_rowlse_kernel, dim3((n1 + 1 + 3) / 4, batch), dim3(256), 0, stream, ot.PT, n1 + 1, n0 + 1,
                       ot.ldpt, ot.v, ot.ldp);
    return hipGetLastError();
}

hipError_t launch_ot_scores(int batch, int n0, int n1, int dual, const OtBuffers& ot, float* scores, hipStream_t stream) {
    hipLaunchKernelGGL(ot_scores_kernel, dim3((n0 + 1 + 3) / 4, batch), dim3(256), 0, stream, ot.P, n0, n1, ot.ldp, ot.u,
                       ot.ldpt, ot.v, dual, scores);
    return hipGetLastError();
}

hipError_t launch_ot_maxima(int batch, int n0, int n1, int dual, const OtBuffers& ot, float* max0, int* arg0,
                            float* max1, int* arg1, hipStream_t stream) {
    hipLaunchKernelGGL(ot_maxima_kernel<0>, dim3((n0 + 3) / 4, batch), dim3(256), 0, stream, ot.P, n0, n1, ot.ldp, ot.u,
                       ot.ldpt, ot.v, ot.ldp, dual, max0, arg0);
    hipLaunchKernelGGL(ot_maxima_kernel<1>, dim3((n1 + 3) / 4, batch), dim3(256), 0, stream, ot.PT, n1, n0, ot.ldpt, ot.v,
                       ot.ldp, ot.u, ot.ldpt, dual, max1, arg1);
    return hipGetLastError();
}

int score_maxima_chunks(int n0) { return (n0 + COL_CHUNK - 1) / COL_CHUNK; }

hipError_t launch_score_maxima(const float* scores, int batch, int n0, int n1, float* max0, int* arg0, float* max1,
                               int* arg1, float* colpart_val, int* colpart_arg, hipStream_t stream) {
    const int chunks = score_maxima_chunks(n0);
    hipLaunchKernelGGL(score_rowmax_kernel, dim3((n0 + 3) / 4, batch), dim3(256), 0, stream, scores, n0, n1, max0, arg0);
    hipLaunchKernelGGL(score_colmax_part_kernel, dim3((n1 + 63) / 64, chunks, batch), dim3(256), 0, stream, scores, n0,
                       n1, chunks, colpart_val, colpart_arg);
    hipLaunchKernelGGL(colmax_combine_kernel, dim3((n1 + 255) / 256, batch), dim3(256), 0, stream, colpart_val,
                       colpart_arg, n1, chunks, max1, arg1);
    return hipGetLastError();
}

hipError_t launch_mutual_matches(int batch, int n0, int n1, const float* max0, const int* arg0, const float* max1,
                                 const int* arg1, float p, int64_t* indices0, int64_t* indices1, float* ms0,
                                 float* ms1, int* range_flag, hipStream_t stream, const RaggedCounts* rc) {
    const int n = n0 > n1 ? n0 : n1;
    RaggedCounts r;
    if (rc) r = *rc; else r.on = 0;
    hipLaunchKernelGGL(mutual_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, stream, n0, n1, max0, arg0, max1, arg1,
                       p, indices0, indices1, ms0, ms1, range_flag, r);
    return hipGetLastError();
}

hipError_t launch_score_mass(const float* scores, int n0, int n1, float* mass0, float* mass1, float* colpart,
                             hipStream_t stream) {
    const int chunks = score_maxima_chunks(n0);
    hipLaunchKernelGGL(score_rowsum_kernel, dim3((n0 + 3) / 4), dim3(256), 0, stream, scores, n0, n1, mass0);
    hipLaunchKernelGGL(score_colsum_part_kernel, dim3((n1 + 63) / 64, chunks), dim3(256), 0, stream, scores, n0, n1,
                       colpart);
    hipLaunchKernelGGL(colsum_combine_kernel, dim3((n1 + 255) / 256), dim3(256), 0, stream, colpart, n1, chunks, mass1);
    return hipGetLastError();
}
