// Internal kernel interface of libimp_hip.so (gfx950 only).  Host launchers live next to their
// kernels in the .hip files; context.hip orchestrates them.  Nothing here is part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IMP_NUM_HEADS 4          // hard-coded in nets/layers.py:157,230

// RAGGED BATCHES (round 4): the pairs of a batch may hold different numbers of keypoints (real SuperPoint output, nets/superpoint.py:204-216;
// the reference therefore runs batch 1, eval/eval_imp.py:60-70).  Tensors stay rectangular - padded to the largest pair, all strides from
// the padded sizes - and every kernel takes the per-pair counts BY VALUE in its parameter block (scalar loads from the kernarg segment: no
// device buffer, nothing to keep alive across asynchronous launches): rows / keys / statistics beyond a pair's own count do not exist for
// it.  A count of 0 retires a pair (every workgroup of it leaves at once): the lock-step loops park finished pairs that way.
#define IMP_RAGGED_MAX 16        // pairs per ragged call (larger batches: several calls)
struct RaggedCounts {
    int on;                      // 0: every pair has the launch's uniform sizes
    int n[2][IMP_RAGGED_MAX];    // [image][pair]
};
#ifdef __HIPCC__
// key split of one (pair, side) unit of the f16x3 ping-pong attention kernel: a function of the unit's OWN query / key counts only (a pair's attention
// output must not depend on the batch it travels in).  Units of up to 4 query tiles (<= 32 workgroups for the pair's 4 heads x 2 sides) are cut in 4,
// up to 6 tiles in 2, larger ones fill the chip well enough alone or with a neighbour; at least 4 key tiles per share
__host__ __device__ __forceinline__ int attn_side_splits(int nq, int nk) {
    if (nq <= 192) return 1;
    const int qt = (nq + 255) / 256;
    int s = qt <= 4 ? 4 : (qt <= 6 ? 2 : 1);
    const int tiles = (nk + 63) / 64;
    while (s > 1 && tiles < 4 * s) --s;
    return s;
}
__device__ __forceinline__ int imp_count(const RaggedCounts& rc, int img, int b, int uniform) { return rc.on ? rc.n[img][b] : uniform; }
#endif

// opt-in to more than 48 KB of dynamic LDS for `kernel`: the attribute is per device, so the largest size already granted is
// tracked per (device, kernel) under a mutex (worker threads launch concurrently; a process may drive several GPUs)
hipError_t imp_grant_dynamic_lds(const void* kernel, size_t bytes);

#ifdef __HIPCC__
// a / b for many numerators over ONE denominator, bit-identical to the IEEE division (round 5): y = 1.0f / b once (a true division), then per numerator
// q0 = a y; r = fma(-q0, b, a) (exact); q = fma(r, y, q0) - Markstein's correction step, correctly rounded when y is the correctly rounded reciprocal and
// nothing over- or underflows (tools/probe/div3_probe.hip: 0 mismatches in 5.5e11 random quotients; the bare product a y differs in 27 % of them).
// 3 VALU instructions instead of the ~10 of v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float imp_div_by(float a, float b, float y) {
    const float q0 = a * y;
    return __builtin_fmaf(__builtin_fmaf(-q0, b, a), y, q0);
}

// POST-MORTEM of a voided waiting launch (round 6, VERDICT r5 #2b).  The mapped host page of a context (context.hip: 64 ints) holds, from word
// IMP_PM_BASE on, one record written by the FIRST waiter whose bounded wait ran out (a system-scope compare-and-swap on word 0 elects it):
//   0 kind (1 resident Sinkhorn wait, 2 Sinkhorn XCC share exceeded, 3 fused layer statistics exchange)   1 the launch's tag (base)
//   2 blockIdx.x   3 HW_ID (wave / SIMD / CU / SH / SE of the waiter)   4 XCC_ID   5 phase (Sinkhorn: 1 partial vectors, 2 half-sum swap, 3 v,
//   4 column maxima; fused layer: 1 block records, 2 finalised statistics)   6 index it waited for (granule / chunk; for kind 2 the slot it drew)
//   7 tag expected (kind 2: the XCC's capacity)   8 tag seen (kind 2: the XCC)   9 iteration   10 pair   11 group / tile   12 G / tiles   13 LOCAL   14 B
// read back - and cleared - by imp_resident_postmortem (include/imp_hip.h); the host adds what it knew when it noticed (words 16..).
#define IMP_PM_BASE 16
__device__ __forceinline__ void imp_postmortem_write(int* host, int kind, unsigned launch_tag, int phase, int idx, unsigned want, unsigned seen, int it, int pair,
                                                     int group, int G, int local, int B) {
    int* pm = host + IMP_PM_BASE;
    int expected = 0;
    if (!__hip_atomic_compare_exchange_strong(pm, &expected, kind, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
    const int vals[14] = {(int)launch_tag, (int)blockIdx.x, (int)__builtin_amdgcn_s_getreg((31 << 11) | 4), (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u),
                          phase, idx, (int)want, (int)seen, it, pair, group, G, local, B};
#pragma unroll
    for (int i = 0; i < 14; ++i) __hip_atomic_store(pm + 1 + i, vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Split-precision operands: x = hi + lo with hi = f16(x) (round to nearest even) and lo = f16(x - hi); x - hi is exact
// in fp32, so lo carries the next 11 significant bits.  Two values at a time: one packed convert for the hi halves and
// one mixed-precision FMA per lo half (v_fma_mix{lo,hi}_f16 reads the f16 hi half and the fp32 x directly and rounds
// x - hi once into the low / high half of the destination) - 3 VALU instructions per pair of values.
typedef _Float16 imp_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void imp_split2(float a, float b, unsigned& hi, unsigned& lo) {
    imp_f16x2 h;
    h[0] = (_Float16)a;
    h[1] = (_Float16)b;
    hi = __builtin_bit_cast(unsigned, h);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(b));
    lo = l;
}
#endif

// ------------------------------------------------------------------------------------------------
// MFMA GEMM (fp32 operands; products as f16x3 split MFMAs or native fp32 MFMAs, see GemmParams::prec):
//   C[z] = epilogue( prologue(A[z]) [M x K] * W[z]^T [K x N] )
// z = (b * nsub + sub) * nside + side ; each side has its own operand set (image 0 / image 1).
// ------------------------------------------------------------------------------------------------
enum {
    GEMM_PRO_NORM = 1 << 0,       // A' = act(norm(A)) per K-channel (InstanceNorm from in_stats, or fixed BN)
    GEMM_PRO_AFFINE = 1 << 1,     // norm has gamma/beta (BatchNorm)
    GEMM_EPI_EXPROW = 1 << 2,     // v = exp(v - rowvec[row])   (probability re-materialisation)
    GEMM_EPI_STATS = 1 << 3,      // per-column partial (sum, sumsq) over the tile's rows -> out_stats
    GEMM_EPI_NOSTORE = 1 << 4,    // do not write C (column statistics only)
    GEMM_EPI_DIV = 1 << 5,        // v = acc / div  (before bias)
};

struct GemmSide {
    const float* A;        // [b][sub][M][lda]
    const float* A2;       // optional second K-range source (k >= ksplit), same strides as A
    const float* W;        // [b][sub][N][ldw]   (weights: strides 0)
    float* C;              // [b][sub][M][ldc]
    const float* R;        // residual, [b][M][ldr] or null
    const float* rowvec;   // [b][sub][M] or null
    const float* in_stats; // [b][K][2] finalised (mean, rstd) of the producer (launch_stats_finalize), or null
    float* out_stats;      // [b][row_tiles][N][2]
    long sA_b, sA_s, sW_b, sW_s, sC_b, sC_s, sR_b, sRV_b, sRV_s;
    int M, N;              // ragged launches (GemmParams::rc): M is the PADDED row count (strides, statistics layout), a pair's own count comes from rc.n[img][b]
    int img;               // image index of this side in GemmParams::rc
};

struct GemmParams {
    GemmSide side[2];
    const float* bias;     // [N] or null
    const float* nm_mean;  // fixed per-channel normalisation (BatchNorm eval) or null
    const float* nm_rstd;
    const float* nm_gamma;
    const float* nm_beta;
    int K, ksplit;         // ksplit: first k served by A2 (== K when unused)
    int lda, lda2, ldw, ldc, ldr;
    int nside, nsub;
    int flags, act;
    int prec;              // 0: fp32 MFMA, 1: f16x3 split (hi/lo halves, 3 f16 MFMAs per fp32 product)
    float div, norm_eps;
    RaggedCounts rc;       // per-pair row counts (rows of A / C; the N dimension is never ragged here)
};

hipError_t launch_gemm_f32(const GemmParams& p, int batch, hipStream_t stream);
// InstanceNorm statistics: per-block (sum, M2 about the block mean) [b][tiles][K][2] of a producer -> (mean, rstd) [b][K][2];
// block t = rows [t * tile_rows, min(M, (t + 1) * tile_rows)); merged with Chan's formula in fp64 in a fixed order, once
// per (batch, side) instead of once per consumer workgroup
struct StatsSide { const float* part; float* out; int tiles; int M; int tile_rows; };      // ragged: tiles / M of the PADDED size (the layout), counts from rc
hipError_t launch_stats_finalize(const StatsSide sides[2], int nside, int batch, int K, float eps, hipStream_t stream, const RaggedCounts* rc = nullptr);
// rows per statistics block of the launch these sizes will get (half the tile height: one wave's rows)
int gemm_stats_rows(int M, int N, int total_z);
int gemm_tile_m(int M, int N, int total_z);

// ------------------------------------------------------------------------------------------------
// attention (flash-style, softmax in registers; launch_attention_f16x3 = split-half MFMA, launch_attention_f32 = fp32 MFMA)
// ------------------------------------------------------------------------------------------------
struct AttnSide {
    const float* q;        // [b][nq][ldq]   head h at columns h*DH
    const float* k;        // [b][nk][ldk]
    const float* v;        // [b][nk][ldk]
    float* out;            // [b][nq][ldo]
    float* lse;            // [b][H][nq] or null
    const uint8_t* kmask;  // [b][nk] or null (1 = key kept)
    long sq_b, sk_b, so_b;
    int nq, nk;            // ragged launches (AttnParams::rc): the padded sizes (lse / mask strides); a pair's own counts come from rc.n[qimg / kimg][b]
    int qimg, kimg;        // image index of the queries / keys in AttnParams::rc
};
struct AttnParams {
    AttnSide side[2];
    int nside, ldq, ldk, ldo, dh;
    // key-split scratch of the f16x3 ping-pong kernel (optional; null = never split): partial results
    // [unit][split][256 x dh + 512] and one zero-initialised, self re-arming ticket per unit, unit = (pair, side, head, query tile)
    float* split_ws;
    unsigned* split_cnt;
    int kv_planes;         // EXPERIMENT (ping-pong kernel only): k / v rows hold, per head, [dh hi halves | dh lo halves] (the split-half image the
                           // kernel otherwise builds while staging) in the bytes of the head's dh floats: staged by plain copy
    RaggedCounts rc;       // per-pair query / key counts (no key mask with it)
    int share_mode;        // who computes the key shares of a split unit: 0 = the launcher decides (attention_f16x3.hip), 1 = one workgroup per share, 2 = one workgroup all of them (tests: same bits)
    // timing hook (imp_time_attention_clock; null in the product path): workgroup 0 adds its lifetime to [0] in shader cycles (s_memtime) and to
    // [1] in ticks of the constant 100 MHz counter (s_memrealtime): [0] / [1] x 100 MHz = the clock the kernel really ran at
    unsigned long long* clk_probe;
};
// in place: the dh-float head segments of columns [col0, col0 + 4 dh) of `rows` rows -> [dh hi halves | dh lo halves]
// ... and back: out[row][h * dh + ch] = hi + lo (exact in fp32)
hipError_t launch_attn_kv_unplanes(const float* base, long rows, int ld, int col0, int dh, float* out, int ldo, hipStream_t stream);
int attention_f16x3_splits(const AttnParams& p, int batch);                        // splits launch_attention_f16x3 will use
size_t attention_f16x3_split_floats(const AttnParams& p, int batch, int nsplit);   // floats of split_ws it needs
size_t attention_f16x3_split_units(const AttnParams& p, int batch);                // tickets it needs
hipError_t launch_attention_f32(const AttnParams& p, int batch, hipStream_t stream);
// split-precision variant (hi/lo halves, 3 f16 MFMAs per fp32 product): attention_f16x3.hip
hipError_t launch_attention_f16x3(const AttnParams& p, int batch, hipStream_t stream);

// column sums of the probability matrix: colsum[b][side][h][key] = sum_q exp(q.k*scale - lse[q])
struct ColsumSide {
    const float* q; const float* k; const float* lse; float* out;   // out [b][H][nk]
    const uint8_t* kmask;                                            // [b][nk] or null: masked keys receive 0
    long sq_b, sk_b;
    int nq, nk;
    int lq;                                                          // stride between the heads of lse (0: nq) - a pair of a padded ragged batch
};
struct ColsumParams { ColsumSide side[2]; int nside, ldq, ldk, dh; };
hipError_t launch_attn_colsum(const ColsumParams& p, int batch, int prec, hipStream_t stream);   // prec: GemmParams::prec

// ------------------------------------------------------------------------------------------------
// keypoint encoder first layer + normalisation
// ------------------------------------------------------------------------------------------------
hipError_t launch_normalize_kpts(const float* kpts, long count, float width, float height, float* out,
                                 hipStream_t stream);
// y[b][n][C0] = W0[C0][3] . (x, y, score) + b0 ; optional fused normalisation (width > 0)
// plus per-channel partial statistics (tiles of 128 tokens) for the following InstanceNorm
struct Kenc0Side { const float* kpts; const float* scores; float* y; float* stats; int n; };      // ragged: n = the padded count
hipError_t launch_kenc_first(const Kenc0Side sides[2], int batch, int c0, const float* W0, const float* b0,
                             float width, float height, hipStream_t stream, const RaggedCounts* rc = nullptr);

// ------------------------------------------------------------------------------------------------
// optimal transport (probability-domain Sinkhorn, nets/layers.py:27-46) + matches
// ------------------------------------------------------------------------------------------------
struct OtBuffers {
    float* P;      // [B][n0+1][ldp]      row softmax of the dustbin-augmented matrix (or logits for dual softmax)
    float* PT;     // [B][n1+1][ldpt]     transpose of P
    float* u;      // [B][n0+1]
    float* v;      // [B][n1+1]   (always the newest v: the fused Sinkhorn path ping-pongs v <-> v2)
    float* v2;     // second v buffer
    float* partials; // [B][ceil((n0+1)/16)][ldp] column partial sums of the fused Sinkhorn pass (null: two-pass path)
    unsigned* P24;   // [B][n0+1][3*ldp/4 dwords] 3-byte copy of P (sign, exponent, 15 mantissa bits, round to nearest even)
                     // read by the fused Sinkhorn iterations instead of P when `compact`; scores / maxima always use P
    int compact;
    int ldp, ldpt;
};
hipError_t launch_ot_init(const float* dist, int batch, int n0, int n1, float bin_score, int dual,
                          const OtBuffers& ot, hipStream_t stream);
hipError_t launch_ot_iterations(int batch, int n0, int n1, int iterations, OtBuffers& ot, hipStream_t stream);
// dual softmax: row / column log-sum-exp into u / v
hipError_t launch_ot_dual_lse(int batch, int n0, int n1, const OtBuffers& ot, hipStream_t stream);
// scores[b][i][j] = (P*u)*v  (or exp(lsr + lsc) for dual)   [B][n0+1][n1+1] contiguous
hipError_t launch_ot_scores(int batch, int n0, int n1, int dual, const OtBuffers& ot, float* scores, hipStream_t stream);
// row / column maxima of the inner block straight from P / PT (fused path, no score tensor)
hipError_t launch_ot_maxima(int batch, int n0, int n1, int dual, const OtBuffers& ot, float* max0, int* arg0,
                            float* max1, int* arg1, hipStream_t stream);
// maxima of an arbitrary score tensor [B][n0+1][n1+1]
hipError_t launch_score_maxima(const float* scores, int batch, int n0, int n1, float* max0, int* arg0,
                               float* max1, int* arg1, float* colpart_val, int* colpart_arg, hipStream_t stream);
int score_maxima_chunks(int n0);
hipError_t launch_mutual_matches(int batch, int n0, int n1, const float* max0, const int* arg0, const float* max1,
                                 const int* arg1, float p, int64_t* indices0, int64_t* indices1, float* ms0,
                                 float* ms1, int* range_flag, hipStream_t stream,    // range_flag (mapped host word or null): set when a maximum is not finite
                                 const RaggedCounts* rc = nullptr);                  // ragged: n0 / n1 = padded sizes; keypoints past a pair's own count get -1 / 0

// ------------------------------------------------------------------------------------------------
// adaptive pooling (nets/adgm.py:552-605) + ragged compaction
// ------------------------------------------------------------------------------------------------
// inner-block row sums / column sums of a score tensor [n0+1][n1+1] (batch element 0)
hipError_t launch_score_mass(const float* scores, int n0, int n1, float* mass0, float* mass1, float* colpart,
                             hipStream_t stream);
// one workgroup per side: threshold, lower medians, union, compaction
struct PoolSide { const float* mass; const float* a_self; const float* a_cross; int64_t* ids; int n; int skip; };
hipError_t launch_pool_select(const PoolSide sides[2], int nsides, float thr, int32_t* counts, hipStream_t stream);
// a[key] = sum_h colsum[h][key] / total   (attention mass received, L1-normalised)
hipError_t launch_attn_mass_normalize(const float* colsum, int n, float* out, hipStream_t stream);
hipError_t launch_gather_rows(const float* in, const int64_t* ids, float* out, int batch, int n_in, int n_out, int dim,
                              hipStream_t stream);
// masked AdaGMN bookkeeping of one pair (pool_misc.hip masked_commit_kernel)
struct MaskedCommit {
    const int64_t *g0, *g1;        // kept ids of image 0 / 1 (unique)
    const int64_t* i0;             // [n0sel] match of kept keypoint t among the kept keypoints of image 1, or -1
    const float* m0;               // [n0sel]
    int64_t* out_i; float* out_m;  // full-size rows of this pair
    int n0sel;
    const int64_t *keep0, *keep1;  // positions selected by the pool, or null (list unchanged)
    int nk0, nk1;
    int64_t *ng0, *ng1;            // composed id lists (null: no update this iteration)
    uint8_t *mask0, *mask1;        // full-size key-mask rows of this pair (zeroed by the caller)
};
hipError_t launch_masked_commit(const MaskedCommit& p, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// chip-resident Sinkhorn: row softmax + T iterations + scores + maxima in ONE launch (ot_resident.hip)
// ------------------------------------------------------------------------------------------------
struct OtResidentParams {
    const float* dist;        // [B][n0][n1]
    int B, n0, n1, T, G;      // G workgroups per pair
    float bin;
    // exchange buffers of 8-byte granules {value, tag}: 2 floats per exchanged float
    float* xpart;             // [B][G][ldx][2]     column partial sums
    float* xv;                // [B][ldx][2]        v (inner columns | dustbin column)
    float* xmax;              // [B][G][2][ldx][2]  column maxima (values | row indices); used only with max0
    unsigned tag_base;        // tags of this launch are tag_base + 1 .. tag_base + 2 T + 1: never reused on these buffers
    int* status;              // device flag, set to 1 when a wait timed out (2: an XCC received more workgroups than its share)
    int* host_status;         // the same word in mapped host memory (read by the library without synchronising), or null
    unsigned* xcc_tickets;    // local != 0: [8] per-XCC ticket counters; a launch adds its per-XCC share to each of them
    unsigned ticket_base;     //   value of the counters before this launch
    unsigned* dev_base;       // launches recorded into a hipGraph: {tag base, ticket base, workgroups done} in DEVICE memory - a replay must not
                              //   reuse its tags, so the launch takes both bases from here and its last workgroup advances them (tag_base /
                              //   ticket_base above are ignored); null for ordinary launches
    int h1;                   // (set by the launcher) LOCAL = 2: workgroups of the second half that take part, 8 .. 32
    int fake_placement;       // TEST HOOK (option ot_fake_placement = 1): workgroups lie about the XCC they run on
    unsigned long long* prof; // optional [6]: phase cycle counts of workgroup 0 (probe), null in the product
    int local;                // 1: XCD-local launch: every pair's G <= 32 workgroups on one XCD (plain stores, L2-served polls);
                              // 2: two XCDs per pair (B <= 4), hierarchical column sums: one fabric crossing per iteration
    float* xhalf;             // local == 2: [2 B][LDX] granules, the half sums the two XCDs of a pair swap
    float* u; int ldu;        // optional outputs in the layout of OtBuffers (u [B][ldu], v [B][ldv]); v is required with u
    float* v; int ldv;
    float* scores;            // optional [B][n0+1][n1+1]
    float* max0; int* arg0;   // optional row / column maxima of the inner block (all four or none)
    float* max1; int* arg1;
    RaggedCounts rc;          // ragged: n0 / n1 above are the PADDED sizes (strides of dist, max0 / max1, u / v); pair b's matrix is rc.n[0][b] x rc.n[1][b];
                              //   no score tensor with it
};
int ot_resident_plan(int batch, int n0, int n1, int max_wgs, int* nch, int* rpw, int* G, int single);   // single: the call holds ONE pair at its own sizes (the wide shapes for n1 > 2048 are allowed)
size_t ot_resident_ldx(int nch);
bool ot_resident_hier_ok(int nch, int rpw, int G, int batch);
hipError_t launch_ot_resident(const OtResidentParams& p, int nch, int rpw, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// weight-fragment split-half GEMM (gemm_wf.hip): static weights pre-split and pre-ordered into MFMA fragment order, the
// activations of a 64-row tile converted once for the whole K extent.  K = 256 or 512, N a multiple of 128.
struct WfSide {
    const float* A;        // [b][M][lda], columns 0 .. ksplit
    const float* A2;       // columns ksplit .. K come from here ([b][M][lda2], its column 0 = k - ksplit), or null
    float* C;              // [b][M][ldc]
    const float* R;        // residual [b][M][ldr] or null
    const float* in_stats; // [b][K][2] finalised (mean, rstd): InstanceNorm + ReLU applied while staging, or null
    float* out_stats;      // [b][row_tiles][N][2] per 64-row block (sum, M2 about the block mean), or null
    float* fin_stats;      // with out_stats: [b][N][2] (mean, rstd), written by the LAST workgroup of the (pair, image) to arrive
                           //   (WfParams::stat_cnt ticket; replaces the stats_finalize launch), or null
    float* C2;             // chained projection output [b][M][ldc2] (WfParams::Wf2_), or null
    long sA_b, sA2_b, sC_b, sR_b, sC2_b;
    int M;                 // ragged launches (WfParams::rc): the PADDED row count (strides, statistics layout); a pair's own count comes from rc.n[img][b]
    int img;               // image index of this side in WfParams::rc
};
struct WfParams {
    WfSide side[2];
    const void* Wf_;       // (u32x4*) fragments of wf_pack
    const float* bias;     // [N] or null
    int K, ksplit, N;
    int lda, lda2, ldc, ldr;
    int nside;
    float norm_eps;        // InstanceNorm eps for fin_stats
    unsigned* stat_cnt;    // [batch][nside] zero-initialised, self re-arming tickets (fin_stats)
    // chained second GEMM on the output tile (K = 512, N = 256 launches with in_stats only): C2 = C_tile . W2^T + bias2, K2 = 256,
    // N2 a multiple of 128 and >= 256 - the next layer's q|k|v (or value) projection of the updated descriptors
    const void* Wf2_;      // fragments of the [N2][256] weights, or null
    const float* bias2;
    int N2, ldc2;
    int kv_image_col, kv_image_col2;   // columns >= this of C / C2 (multiples of 128; INT_MAX-like = none) are written as the split-half image the
                           // attention kernel stages by plain copy: per 64-channel head segment [64 hi halves | 64 lo halves] (AttnParams::kv_planes)
    int pass_split;        // > 1: the N / 128 column passes of a row tile are dealt to this many workgroups (must divide N / 128; not with Wf2_)
    int dbg;               // probe switches (tools/probe/gemm_wf_time.py): 1 no global stores, 2 no epilogue at all, 4 no residual / bias loads
    RaggedCounts rc;       // per-pair row counts
};
hipError_t launch_gemm_wf(const WfParams& p, int batch, hipStream_t stream);
// FUSED layer MLP (round 4): mlp.0 on cat[x, attention output] -> InstanceNorm statistics over ALL keypoints of the image (exchanged
// between the workgroups of the launch) -> ReLU -> mlp.3 + bias + residual (-> the chained projection of WfParams::Wf2_) in ONE launch,
// one workgroup per 64-row tile; the hidden tensor [M][512] never leaves the chip.  WfParams describes the first convolution
// (K = 512, N = 512, side.A / A2 = x / attention output, side.C / R = new / old descriptors, side.C2 = chained output) and this
// struct the rest.  The workgroups of one (pair, image) WAIT for each other: all of them must be co-resident (at most one tile per CU)
// and no other waiting kernel may be dispatched beside it (context.hip SpinGate).
struct WfFused {
    const void* Wf3_;      // fragments of mlp.3 [256][512] (wf_pack)
    const float* bias3;    // [256]
    float* rec[2];         // per image: [b][tiles of the image][512] statistics granules {sum, tag, M2, tag} (16 bytes) of each 64-row block
    float* fin[2];         // per image: [b][512] granules {mean, tag, rstd, tag}
    unsigned tag;          // unique per launch on these buffers, never 0
    int* status;           // device word: set to 3 when a wait timed out (every waiter of the launch then falls through and poisons its rows)
    int* host_status;      // the context's mapped health word (context.hip resident_health), or null
    int fake;              // TEST HOOK: workgroup 0 withholds its statistics - every wait on them times out
    unsigned long long* prof;   // optional [grid][6] phase cycle stamps (WF_PROFILE builds), else null
};
hipError_t launch_gemm_wf_fused(const WfParams& p, const WfFused& f, int batch, hipStream_t stream);
bool gemm_wf_supported(int K, int N);
int gemm_wf_stats_rows();
constexpr int WF_MAX_PSPLIT = 8;      // WfParams::stat_cnt holds batch x nside x WF_MAX_PSPLIT tickets (one per pass group)
// W [N][K] fp32 -> split-half MFMA fragments, 2 N K halves (host)
void wf_pack(const float* W, int N, int K, _Float16* out);
