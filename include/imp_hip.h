/*
 * imp_hip.h -- C-ABI of the MI355X (gfx950) implementation of the IMP / EIMP matching hot path.
 *
 * The reference (feixue94/imp-release) has no FFI layer: its boundary is the Python nn.Module
 * surface of GM / DGNNS / AdaGMN.  This library sits directly below that
 * surface; each entry point replaces one reference method (cited per function, paths relative to
 * the reference root) and is what a ctypes / cffi / pybind stub in the reference would bind
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.  `stream` is a hipStream_t passed
 *    as void* (NULL = default stream).  All launches are asynchronous on that stream; no entry
 *    point synchronises the host unless its comment says so.
 *  - All tensors are DEVICE pointers to float32 unless stated, TOKEN-MAJOR and contiguous:
 *      descriptors / encodings  [B][N][D]      (the reference keeps [B][D][N]; the host mirror
 *                                               hands out transposed views)
 *      keypoints                [B][N][2]       scores [B][N]
 *      dist                     [B][N0][N1]     score matrix [B][N0+1][N1+1]
 *      indices                  int64 [B][N]    (-1 = unmatched, as nets/gm.py:317-318)
 *    "side 0" has n0 keypoints per image, "side 1" has n1; both are uniform over the batch B
 *    (exactly as the reference's rectangular [B,N,*] tensors).
 *  - Return value: 0 = ok, <0 = error (IMP_E_*); message via imp_last_error().  No exceptions,
 *    no caller-visible allocation.  A context is bound to one device, is NOT thread-safe, and -
 *    like the reference modules (nets/gm.py:79-82, nets/layers.py:132,209,216) - is stateful:
 *    it caches the last self / cross attention operands for the attention-sharing layers and
 *    for pooling.  One in-flight pair batch per context.
 *  - Arithmetic: float32 storage and accumulation end-to-end; GEMM-shaped products on the matrix pipe either as
 *    split-half f16x3 MFMAs (default, fp32-level accuracy) or native fp32 MFMAs (imp_set_precision); int64 indices.
 */
#ifndef IMP_HIP_H
#define IMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMP_OK 0
#define IMP_E_ARG (-1)      /* bad argument / shape */
#define IMP_E_STATE (-2)    /* call order (weights not finalised, no cached attention, ...) */
#define IMP_E_HIP (-3)      /* HIP runtime error */
#define IMP_E_NOMEM (-4)
#define IMP_E_KEY (-5)      /* unknown / missing state_dict key */
#define IMP_E_RANGE (-7)    /* an EARLIER call on this context produced non-finite match scores: in the default split-half f16x3 arithmetic every
                             * matrix operand (descriptors, activations, projections) must satisfy |x| < 65504 (fp16 range of the high half);
                             * beyond it - or with non-finite inputs - the scores are NaN and that call's matches are all -1.  Raised by the
                             * match kernel through a mapped host word and reported at the next entry; precision f32 has no such limit */
#define IMP_E_NOFIT (-8)    /* a RAGGED batch (imp_set_counts) that the chip-resident Sinkhorn kernel cannot hold - sizes, or the context has fallen back to
                             * the streaming kernels, which take uniform batches only: nothing was computed; run the pairs in smaller groups (a single
                             * pair whose counts equal the padded sizes always runs: it is a uniform batch) */
#define IMP_E_RESIDENT (-6) /* a chip-resident Sinkhorn launch of an EARLIER call on this context timed out: that call's results are void
                             * (poisoned: mscores NaN, indices -1); the context has recovered on a safer protocol - re-run the batch */

/* model flavours: which GNN layers re-use the previous iteration's attention
 * (nets/gm.py:62 GM = none; nets/gms.py:17 DGNNS and nets/adgm.py:18 AdaGMN = [F,F]*2+[F,F,T,T]*21) */
#define IMP_MODEL_GM 0
#define IMP_MODEL_DGNNS 1
#define IMP_MODEL_ADAGMN 2

#define IMP_NORM_IN 0       /* nn.InstanceNorm1d(eps=1e-3)  nets/layers.py:67-68 */
#define IMP_NORM_BN 1       /* nn.BatchNorm1d(eps=1e-3), eval mode  nets/layers.py:69-70 */
#define IMP_ACT_RELU 0      /* nets/layers.py:71-76 */
#define IMP_ACT_GELU 1
#define IMP_ACT_LRELU 2

typedef struct imp_ctx imp_ctx;

typedef struct imp_config {
    int32_t model;            /* IMP_MODEL_* */
    int32_t descriptor_dim;   /* 256 (SuperPoint) or 128 (SIFT); nets/gm.py:31, eval/eval_imp.py:260 */
    int32_t n_gnn_layers;     /* len(config['GNN_layers']); layer i is 'self' if i even else 'cross' */
    int32_t n_layers;         /* number of final_proj heads (config['n_layers']) */
    int32_t kenc_channels[8]; /* config['keypoint_encoder'], zero-terminated, e.g. {32,64,128,256,0} */
    int32_t norm_fn;          /* IMP_NORM_* */
    int32_t ac_fn;            /* IMP_ACT_* */
    int32_t max_batch;        /* workspace sizing: pairs per call */
    int32_t max_keypoints;    /* workspace sizing: max(n0, n1) */
    int32_t layer_is_cross[128]; /* per GNN layer: 0 = 'self', 1 = 'cross' (config['GNN_layers']) */
} imp_config;

const char* imp_last_error(void);
/* library / build identification, e.g. "imp_hip 0.1 gfx950 f32-mfma" */
const char* imp_version(void);

/* GM.__init__ (nets/gm.py:46-82) + .cuda(): allocates packed-weight storage and workspace on `device`. */
int imp_create(imp_ctx** out, const imp_config* cfg, int device);
int imp_destroy(imp_ctx* ctx);

/* load_state_dict(strict=True) (eval/eval_imp.py:333): one call per tensor, `key` is the reference
 * state_dict key ("kenc.encoder.0.weight", "gnn.layers.3.attn.proj.1.bias", "bin_score", ...),
 * `data` is a HOST pointer to float32 in the reference's own layout ([out][in][1] for conv weights).
 * imp_finalize_weights() checks that every key of the schema was supplied (IMP_E_KEY otherwise),
 * and packs: attention projections are concatenated to one [3D][D] matrix with rows permuted from the
 * reference's interleaved head layout (channel = d*4 + h, nets/layers.py:119-120) to head-major. */
int imp_load_tensor(imp_ctx* ctx, const char* key, const float* data, const int64_t* shape, int ndim);
int imp_finalize_weights(imp_ctx* ctx);
/* Matrix arithmetic of every GEMM-shaped op (projections, attention, score matrix):
 *   1 (default) "f16x3": each fp32 operand is split into two halves hi = f16(x), lo = f16(x - hi) and each fp32
 *       product is three f16 MFMAs with fp32 accumulation - fp32-level results (parity suite green) at 3/16 of
 *       the fp32-MFMA pipe time;
 *   0 "f32": native fp32-input MFMA (v_mfma_f32_32x32x2_f32).
 * Environment override at imp_create: IMP_PRECISION=f32|f16x3. */
int imp_set_precision(imp_ctx* ctx, int precision);
int imp_get_precision(imp_ctx* ctx);
/* Storage the T Sinkhorn iterations (nets/layers.py:31-33) stream: 4 (default) = the fp32 matrix; 3 = a 3-byte copy
 * (sign, exponent, 15 mantissa bits, round to nearest even) - 3/4 of the bytes of the HBM-bound loop.  The returned
 * scores p.u.v and the match maxima are formed from the fp32 matrix either way; with 3, u and v (and so every score)
 * move by ~1e-5 RELATIVE (measured: match scores <= 1.2e-5 from the reference, indices identical), i.e. O(N)-sized
 * dustbin entries move by ~1e-5 N. */
int imp_set_sinkhorn_storage(imp_ctx* ctx, int bytes_per_element);

/* number of schema keys / i-th key (so a host can enumerate what strict loading expects) */
int imp_num_keys(imp_ctx* ctx);
const char* imp_key_name(imp_ctx* ctx, int i);

/* normalize_keypoints (nets/layers.py:49-56): (kpts - [w,h]/2) / (0.7*max(w,h)); kpts,out [B][n][2].
   ctx may be NULL (the reference's free function has no model): the launch then goes to the device owning kpts */
int imp_normalize_keypoints(imp_ctx* ctx, const float* kpts, int batch, int n, float width, float height,
                            float* out, void* stream);

/* GM.encode_keypoint / KeypointEncoder.forward (nets/gm.py:287-288, nets/layers.py:80-90).
 * enc = MLP(cat(norm_kpts, scores)); if `desc*` is non-NULL the residual add of nets/gm.py:177-178 is
 * fused: out = desc + enc.  out* [B][n*][D]. */
int imp_encode_keypoints(imp_ctx* ctx, int batch, int n0, int n1,
                         const float* nkpts0, const float* scores0, const float* desc0, float* out0,
                         const float* nkpts1, const float* scores1, const float* desc1, float* out1,
                         void* stream);

/* GM/DGNNS/AdaGMN.forward_one_layer (nets/gm.py:263-285, nets/gms.py:260-282, nets/adgm.py:528-550):
 * one (Shared)AttentionalPropagation layer on both images, both deltas from the pre-update
 * descriptors, out = desc + delta (out may alias desc).  Shared layers (DGNNS/AdaGMN) re-use the
 * attention of the previous layer of the same kind, which this context caches as (Q, K, row
 * log-sum-exp) rather than as the [B][4][N][M] probability tensor of nets/layers.py:132.
 * key_mask0/1: optional uint8 [B][n] "key is kept" masks for the masked attention of
 * AdaGMN.produce_matches (nets/adgm.py:377-396, nets/layers.py:124-127): key_mask0 masks the
 * image-0 keypoints wherever they act as keys, key_mask1 the image-1 keypoints. */
int imp_forward_layer(imp_ctx* ctx, int layer_i, int batch, int n0, int n1,
                      const float* desc0, const float* desc1, float* out0, float* out1,
                      const uint8_t* key_mask0, const uint8_t* key_mask1, void* stream);

/* Materialise a cached attention probability tensor (what the reference keeps in
 * model.self_prob0/1, model.cross_prob0/1, nets/gm.py:272-283): which = 0 self image0 [B][4][n0][n0],
 * 1 self image1 [B][4][n1][n1], 2 cross image0<-image1 [B][4][n0][n1] (reference "cross_prob1"),
 * 3 cross image1<-image0 [B][4][n1][n0] (reference "cross_prob0"). */
int imp_attention_prob(imp_ctx* ctx, int which, float* prob, void* stream);

/* Attention mass received per key, summed over heads and queries and L1-normalised per batch element
 * (nets/adgm.py:424-432 / 557-565) for one cached probability matrix (`which` as above); out [B][nk]. */
int imp_attention_received(imp_ctx* ctx, int which, float* out, void* stream);

/* GM.compute_distance (nets/gm.py:290-295): final_proj[layer_id] on both sides, dot / sqrt(D). */
int imp_compute_distance(imp_ctx* ctx, int layer_id, int batch, int n0, int n1,
                         const float* desc0, const float* desc1, float* dist, void* stream);

/* GM.compute_score (nets/gm.py:297-303) -> sink_algorithm (nets/layers.py:38-46, 27-35) or
 * dual_softmax (nets/layers.py:20-24).  dist [B][n0][n1]; scores [B][n0+1][n1+1] (may be NULL when
 * only the fused match extraction below is wanted).  `bin_score` < 0 is NOT special: pass the value. */
int imp_compute_score(imp_ctx* ctx, int batch, int n0, int n1, const float* dist, float bin_score,
                      int iterations, int with_sinkhorn, float* scores, void* stream);

/* GM.compute_matches (nets/gm.py:305-320) on an arbitrary score tensor [B][n0+1][n1+1]. */
int imp_compute_matches(imp_ctx* ctx, int batch, int n0, int n1, const float* scores, float p,
                        int64_t* indices0, int64_t* indices1, float* mscores0, float* mscores1, void* stream);

/* AdaGMN.pool (nets/adgm.py:552-605) for batch element 0 using the cached attention of the last
 * self and cross layers.  ids0/ids1: int64 [n0]/[n1] output buffers (ascending kept ids);
 * counts: int32[4] DEVICE buffer = {n_keep0, n_confident0, n_keep1, n_confident1}; a side whose
 * n_confident is 0 or that is skipped by n_min_tokens reports n_keep = -1 (reference: None). */
int imp_pool(imp_ctx* ctx, int n0, int n1, const float* scores, float mscore_th, float uncertainty_ratio,
             int n_min_tokens, int64_t* ids0, int64_t* ids1, int32_t* counts, void* stream);

/* Matching confidence per keypoint used by the pooling (nets/adgm.py:476-477,488-489): row sums and column
 * sums of the inner n0 x n1 block of one score matrix [n0+1][n1+1]; mass0 [n0], mass1 [n1]. */
int imp_score_mass(imp_ctx* ctx, int n0, int n1, const float* scores, float* mass0, float* mass1, void* stream);

/* One side of the pooling selection on caller-supplied vectors of length n (the masked variant of
 * AdaGMN.produce_matches, nets/adgm.py:475-497, works on gathered subsets): confident = mass >= thr;
 * keep = confident | a_self >= lower_median(a_self[confident]) | a_cross >= lower_median(a_cross[confident]).
 * ids int64 [n]; counts int32[2] DEVICE = {n_keep or -1 when nothing is confident, n_confident}. */
int imp_pool_select(imp_ctx* ctx, int n, const float* mass, const float* a_self, const float* a_cross, float thr,
                    int64_t* ids, int32_t* counts, void* stream);
/* Both images of a pair in ONE launch (one count read-back per pair and updating iteration instead of two): side s is skipped
 * (counts[2 s] = -1, nets/adgm.py:465-473: no more than n_min_tokens keypoints left) when skip_s != 0.  counts int32[4] DEVICE =
 * {n_keep0 | -1, n_confident0, n_keep1 | -1, n_confident1}. */
int imp_pool_select_pair(imp_ctx* ctx, int n0, const float* mass0, const float* a_self0, const float* a_cross0, int skip0, int64_t* ids0,
                         int n1, const float* mass1, const float* a_self1, const float* a_cross1, int skip1, int64_t* ids1, float thr,
                         int32_t* counts, void* stream);

/* Masked AdaGMN.produce_matches, bookkeeping of ONE pair in one launch (nets/adgm.py:447-453: batch_indices0[bi, gids0[valid0]] =
 * gids1[indices0[valid0]], batch_mscores0[bi, gids0] = mscores0; :498-504: the kept id lists and the masks M00/M01/M11/M10 of the next
 * layers).  gids0 [n0sel] / gids1: kept ids (unique); indices0 / mscores0 [n0sel]: matches among the kept keypoints; out_* : this
 * pair's full-size rows (pre-filled -1 / 0).  With new_gids0/1 != NULL the lists are composed with the pool's selection
 * (new_gids_s[t] = keep_s ? gids_s[keep_s[t]] : gids_s[t], t < nkeep_s; keep_s NULL: list unchanged, nkeep_s = its length) and
 * mask_s[new id] = 1 (uint8 rows of this pair, zeroed by the caller).  No synchronisation, no read-back. */
int imp_masked_commit(imp_ctx* ctx, int n0sel, const int64_t* gids0, const int64_t* gids1, const int64_t* indices0, const float* mscores0,
                      int64_t* out_indices0, float* out_mscores0, const int64_t* keep0, int nkeep0, const int64_t* keep1, int nkeep1,
                      int64_t* new_gids0, int64_t* new_gids1, uint8_t* mask0, uint8_t* mask1, void* stream);

/* desc[:, :, sel_ids] of eval/matching.py:166-174 on token-major data: out[b][i][:] = in[b][ids[i]][:] */
int imp_gather_rows(imp_ctx* ctx, int batch, int n_in, int n_out, int dim, const float* in,
                    const int64_t* ids, float* out, void* stream);

/* Fused one-shot matcher = GM/DGNNS.produce_matches(data, p, only_last=True)
 * (nets/gm.py:145-247, nets/gms.py:139-258): normalise (if width>0, else kpts are already
 * normalised), encode, all GNN layers, final_proj[n_layers-1], Sinkhorn / dual-softmax, mutual matches.
 * Nothing is synchronised or allocated once the workspace is sized, so the call can be recorded into a hipGraph (stream capture) and
 * replayed: the chip-resident Sinkhorn launch is recorded too (its exchange tags live in device memory and advance with every
 * replay; option ot_graph = 0: the streaming kernels instead).  Such a graph must not be replayed while another resident launch of the
 * process runs - a collision is reported like any voided resident launch (NaN scores, IMP_E_RESIDENT at the next entry point).
 * A graph records the exchange protocol the context used at capture time: after an IMP_E_RESIDENT the context has stepped down to a
 * safer one, and graphs captured before that must be captured again (replaying them would time out again on every replay).
 * Outputs: indices0 int64 [B][n0], mscores0 [B][n0] (+ optional indices1,
 * mscores1, scores [B][n0+1][n1+1], any of which may be NULL). */
int imp_match_pair(imp_ctx* ctx, int batch, int n0, int n1,
                   const float* kpts0, const float* scores0, const float* desc0,
                   const float* kpts1, const float* scores1, const float* desc1,
                   float width, float height, float bin_score, int sinkhorn_iterations, int with_sinkhorn,
                   float p, int64_t* indices0, float* mscores0, int64_t* indices1, float* mscores1,
                   float* scores, void* stream);

/* ---- op-level entry points (used by the parity tests and by bench.py's roofline leg) ---------- */

/* y[M][N] = x[M][K] @ W[N][K]^T + bias   (fp32 MFMA GEMM that every 1x1 conv maps to) */
int imp_op_linear(imp_ctx* ctx, int M, int N, int K, const float* x, const float* W, const float* bias,
                  float* y, void* stream);
/* test entry of the weight-fragment layer GEMM (csrc/gemm_wf.hip: the kernel behind the three 1x1 convolutions of a GNN layer,
 * nets/layers.py:119-120,145-149,210-218), one image side, B batch elements, device pointers, row-major:
 *   h = x, or - stats_in [B][ksplit][2] (mean, rstd per input channel) given - relu((x - mean) * rstd)        (InstanceNorm + ReLU)
 *   y [B][M][N] = [h | x2] @ W[N][K]^T + bias (+ residual [B][M][N]);  x [B][M][ksplit], x2 [B][M][K - ksplit] (ksplit == K: x2 unused)
 *   stats_out (optional) [B][N][2]: (mean, 1 / sqrt(biased var + 1e-3)) of y over the M rows of each batch element - the per-block
 *     statistics epilogue + the last-arrival merge inside the launch
 *   W2 [N2][256], bias2, y2 [B][M][N2] (optional; K = 512, N = 256, stats_in): y2 = y @ W2^T + bias2 on the tile still in LDS
 *   pass_split: column passes of a row tile dealt to this many workgroups (1 = none; must divide N / 128)
 * K = 256 or 512, N % 128 == 0.  Packs the weights on the fly; synchronises. */
int imp_op_layer_gemm(imp_ctx* ctx, int B, int M, int N, int K, int ksplit, const float* x, const float* x2, const float* W,
                      const float* bias, const float* residual, const float* stats_in, float* y, float* stats_out,
                      const float* W2, const float* bias2, int N2, float* y2, int pass_split, void* stream);
/* multi-head attention core on packed projections: qkv_q [B][nq][3D], qkv_kv [B][nk][3D]
 * (q | k | v, head-major), out [B][nq][D], lse [B][4][nq] (optional).  nets/layers.py:121-131 */
/* RAGGED BATCHES (round 4).  Real SuperPoint output holds a different number of keypoints per image (nets/superpoint.py:204-216:
 * threshold, border filter, top-k), which is why the reference's drivers run one pair at a time (eval/eval_imp.py:60-70).  Here a batch
 * may be ragged: the tensors stay rectangular, padded to the largest pair of the batch (every n0 / n1 argument below = the padded size),
 * and imp_set_counts gives the per-pair counts that the NEXT calls on this context obey - imp_encode_keypoints, imp_forward_layer,
 * imp_compute_distance, imp_match_pair, imp_match_tail: keypoints / keys / InstanceNorm statistics past a pair's own count do not
 * exist for it, its result equals the pair run alone; outputs past the count are -1 (indices) / 0 (scores) or unspecified
 * (descriptors).  Not with key masks, a score tensor or the dual-softmax scorer.  A count of 0 for BOTH images retires a pair: its
 * workgroups leave at once (the lock-step loops of imp_release_amd.eval_loop park finished pairs that way).  batch <= 16; the counts
 * are HOST arrays, copied; n0 = n1 = NULL returns the context to uniform batches. */
int imp_set_counts(imp_ctx* ctx, int batch, const int32_t* n0, const int32_t* n1);
/* the tail of imp_match_pair on its own, for loops that score at several iterations (eval/matching.py:55-61: compute_distance
 * (final_proj[layer_id], nets/gm.py:290-295) -> compute_score -> compute_matches) without materialising the score tensor; obeys
 * imp_set_counts */
int imp_match_tail(imp_ctx* ctx, int layer_id, int batch, int n0, int n1, const float* desc0, const float* desc1, float bin_score,
                   int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0, float* mscores0, int64_t* indices1,
                   float* mscores1, void* stream);
/* the same with the score tensor of the Sinkhorn scorer as a by-product (what AdaGMN.pool consumes, nets/adgm.py:552-605): scores holds
 * batch slots of (n0 + 1) * (n1 + 1) floats (n0, n1 = the padded sizes); pair b's tensor lies at the start of slot b, DENSE with the pair's
 * own shape [m0 + 1][m1 + 1] (m = its counts, imp_set_counts; uniform batches: m = n) - the tensor the pair run alone returns from
 * imp_compute_score.  scores = NULL: imp_match_tail. */
int imp_match_tail_scores(imp_ctx* ctx, int layer_id, int batch, int n0, int n1, const float* desc0, const float* desc1, float bin_score,
                          int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0, float* mscores0, int64_t* indices1,
                          float* mscores1, float* scores, void* stream);
/* imp_pool for pair `pair` of the batch whose attention is cached (batch, n0, n1 = its padded shape; the pair's own sizes from
 * imp_set_counts): scores = that pair's dense score tensor (slot `pair` of imp_match_tail_scores); ids0 / ids1 / counts as imp_pool.  The
 * result equals imp_pool of the pair run alone (round 4: EIMP pairs of different sizes advance in lock step, eval/matching.py:126-276). */
int imp_pool_pair(imp_ctx* ctx, int pair, int batch, int n0, int n1, const float* scores, float mscore_th, float uncertainty_ratio,
                  int n_min_tokens, int64_t* ids0, int64_t* ids1, int32_t* counts, void* stream);
/* The lock-step IMP loop (eval/matching.py:16-123 `matching_iterative` on B pairs of different sizes at once) driven natively: encoder, per iteration
 * the two layers (chained projections), at the iterations of `valid_mask` (bit it: eval/matching.py:43 = {3,5,7,9,11,13,14}) final projection
 * -> distance -> Sinkhorn -> mutual matches at `match_ratio` for the whole ragged batch, one device->host copy, and per live pair the
 * reference's host step: matched-count gate (`min_kpts`), pose estimate (imp_estimate_pose on `pose_threads` worker threads of the context, each
 * with its own stream; 0 = no pose step: no pair ever exits), pose-change test against `stop_pose_deg` (< 0: never stop).  The next iteration's
 * layers are enqueued BEFORE the host turns to the pose step and the exit decision of a scored iteration is taken at the next scored one - the
 * pose step is latency, not GPU load; a pair that exits is returned with the matches (inlier-filtered, :112-113), pose and iteration count of
 * the iteration it exited at, exactly as the sequential loop returns them, and is then RETIRED (count 0).  Pairs that never exit get the
 * p = 0.2 matches of the last scored iteration (:119; derived from the scored matches: mscores0 do not depend on the threshold).
 * Inputs as for imp_encode_keypoints (normalised keypoints, device tensors padded to n0 / n1) + per-pair host records.  Synchronises.
 * IMP_E_RESIDENT: a waiting kernel of the group was voided - run the group again.  (The Python twin, with any estimate_pose callable:
 * imp_release_amd.matching.matching_iterative_lockstep.) */
typedef struct imp_loop_pair {
    const float* pts0;      /* [n0[b]][2] pixel keypoints of image 0 (host): what the pose step sees (eval/matching.py:84-87) */
    const float* pts1;      /* [n1[b]][2] */
    const double* K0;       /* 3x3 row-major intrinsics (host) */
    const double* K1;
    int64_t* indices0;      /* out [n0[b]] (host) */
    float* mscores0;        /* out [n0[b]] (host) */
    double R[9];            /* out: pose of the exit iteration (found = 1) */
    double t[3];
    int32_t found;          /* out: 1 = the pair left the loop on pose convergence */
    int32_t n_iterations;   /* out: eval/matching.py's n_iter */
} imp_loop_pair;
int imp_loop_lockstep(imp_ctx* ctx, int batch, const int32_t* n0_counts, const int32_t* n1_counts, int n0, int n1, const float* nkpts0,
                      const float* scores0, const float* desc0, const float* nkpts1, const float* scores1, const float* desc1, float bin_score,
                      int sinkhorn_iterations, int n_iterations, unsigned valid_mask, float match_ratio, int min_kpts, double error_th,
                      double stop_pose_deg, int pose_threads, int pose_iterations, unsigned pose_seed, int pose_flags, imp_loop_pair* pairs,
                      void* stream);

/* The EIMP loop (eval/matching.py:126-276: adaptive pooling between the iterations) on `batch` pairs in lock step, host logic included.
 * As imp_loop_lockstep, plus: after every scored iteration each live pair is pooled on its own slice of the ragged batch (imp_pool_pair
 * semantics; threshold 0.2, or - with_uncertainty - 0.2 x the inlier ratio of the pair's pose estimate, :243-247; n_min_tokens as
 * AdaGMN.pool, the reference loop passes its default 256) and the next iteration runs on the kept rows.  The pose estimates of an
 * iteration run side by side on the context's worker threads; the group waits for all of them (the pool threshold needs them).
 * Per pair out: kept0 / kept1 = indices of the surviving keypoints in the pair's own numbering (ascending), indices0 / mscores0 = the
 * matches among them (n_indices entries; = n_kept0 unless the loop ends on an unscored iteration after a pool).
 * (Python twin: imp_release_amd.matching.matching_iterative_uncertainty_lockstep.) */
typedef struct imp_loop_pair_u {
    const float* pts0;      /* [n0[b]][2] pixel keypoints of image 0 (host) */
    const float* pts1;      /* [n1[b]][2] */
    const double* K0;       /* 3x3 row-major intrinsics (host) */
    const double* K1;
    int64_t* indices0;      /* out, capacity n0[b] (host) */
    float* mscores0;        /* out, capacity n0[b] (host) */
    int32_t* kept0;         /* out, capacity n0[b] (host) */
    int32_t* kept1;         /* out, capacity n1[b] (host) */
    double R[9];            /* out: pose of the exit iteration (found = 1) */
    double t[3];
    int32_t found;          /* out: 1 = the pair left the loop on pose convergence */
    int32_t n_iterations;   /* out: eval/matching.py's n_iter */
    int32_t n_kept0, n_kept1, n_indices, reserved;
} imp_loop_pair_u;
int imp_loop_lockstep_uncertainty(imp_ctx* ctx, int batch, const int32_t* n0_counts, const int32_t* n1_counts, int n0, int n1,
                                  const float* nkpts0, const float* scores0, const float* desc0, const float* nkpts1, const float* scores1,
                                  const float* desc1, float bin_score, int sinkhorn_iterations, int n_iterations, unsigned valid_mask,
                                  float match_ratio, int min_kpts, double error_th, double stop_pose_deg, int with_uncertainty,
                                  int n_min_tokens, int pose_threads, int pose_iterations, unsigned pose_seed, int pose_flags,
                                  imp_loop_pair_u* pairs, void* stream);
/* the fused layer MLP of csrc/gemm_wf.hip on its own (tests/test_gpu_ops.py): nets/layers.py:145-149 / :210-218 after the attention,
 *   y = x + mlp.3(relu(InstanceNorm(mlp.0(cat[x, a]))))        x, a, y: [B][M][256];  W0 [512][512], W3 [256][512]
 *   y2 = y . W2^T + b2                                         optional (W2 [N2][256], N2 % 128 == 0): the next layer's projection
 * in ONE launch whose B * ceil(M / 64) workgroups exchange the InstanceNorm statistics inside the kernel (must not exceed the CU
 * count).  fake != 0: test hook, one workgroup withholds its statistics -> the exchange times out (seconds), outputs NaN, IMP_E_RESIDENT.
 * Synchronises. */
int imp_op_fused_mlp(imp_ctx* ctx, int B, int M, const float* x, const float* a, const float* W0, const float* b0, const float* W3,
                     const float* b3, const float* W2, const float* b2, int N2, float* y, float* y2, int fake, void* stream);
int imp_op_attention(imp_ctx* ctx, int batch, int nq, int nk, int dim, const float* qkv_q,
                     const float* qkv_kv, const uint8_t* key_mask, float* out, float* lse, void* stream);
/* Named switches of a context, to be set before its first compute call; not part of the reference's operator interface (the reference's eval scripts know nothing like it).  The
 * paths they select are the ones a context steps down to BY ITSELF after a voided waiting launch - forced here for A/B tests - and the test suite's fault hooks:
 *   ot_resident 0|1   chip-resident Sinkhorn kernel (0: streaming kernels)          ot_local / ot_hier 0|1   its XCD-local / two-XCD decompositions
 *   ot_verify 0|1     wait for every resident launch and repair it inside the call   ot_graph 0|1             0: hipGraph captures record the streaming Sinkhorn
 *   gemm_wf 0|1|2     weight-fragment GEMMs (0: plain tiled GEMMs, 2: at every size)  wf_chain 0|1, wf_chain_min   MLP3 + next projection in one launch
 *   wf_fused 0|1|2, wf_fused_min   the layer's MLP in one launch with the in-kernel InstanceNorm statistics exchange (1: while the stream has the chip, 2: always)
 *   fused_choice 0|1  when a stream counts as alone and takes the fused layer launch: 0 after seven kernel choices in a row (default), 1 after 8 ms (A/B; process-wide);
 *                     fused_alone_after K: the count of that rule (default 6: more than six in a row)
 *   attn_shares 0|1|2 key shares of a split attention unit by one workgroup each (1) or all by one (2); 0: the launcher decides - same bits either way
 *   kv_image 0|1      k | v written as split-half images by the projection          probe_prof 0|1           timing entry points print a profiling build's stamps
 *   ot_fake_placement, wf_fused_fake, ot_graph_tag0       fault injection (tests/test_gpu_resident_ot.py, tests/test_gpu_parity.py)
 * The environment knows three switches only - IMP_PRECISION=f32, IMP_OT_RESIDENT=0, IMP_WF_FUSED=0 (a process that shares the GPU turns the two waiting kernels
 * off) - plus IMP_OPTIONS="name=value,..." which applies a list of the above to every context the process creates.  Unknown name: IMP_E_ARG. */
int imp_ctx_option(imp_ctx* ctx, const char* name, long value);
/* timing hooks for bench.py: hipEvent-bracketed repetition of one attention / one Sinkhorn pass on
 * the context's own stream-ordered workspace; returns average milliseconds per launch in *ms. */
int imp_time_attention(imp_ctx* ctx, int batch, int n, int reps, float* ms, void* stream);
/* the same launches with the clock they ran at: workgroup 0 of every timed launch reads the shader-cycle counter (s_memtime) and the constant
 * 100 MHz counter (s_memrealtime) on entry and exit; *sclk_mhz = cycles / ticks x 100 (0 when the kernel variant taken carries no probe).
 * bench.py reports it as roofline.sclk_mhz_observed: the MFMA roof at THAT clock is what the launch could have reached. */
int imp_time_attention_clock(imp_ctx* ctx, int batch, int n, int reps, float* ms, float* sclk_mhz, void* stream);
/* same for the Sinkhorn iterations (nets/layers.py:31-33) over [batch][n+1][n+1] on the path the product takes for that
 * shape (chip-resident kernel: time(T iterations) - time(0 iterations); streaming path: 2 launches per iteration);
 * *ms = average milliseconds per ITERATION */
int imp_time_sinkhorn(imp_ctx* ctx, int batch, int n, int iterations, float* ms, void* stream);
/* probe (tools/probe/gemm_time.py): average milliseconds of one of the three GEMMs of GNN layer 0 (which: 0 q|k|v projection,
 * 1 MLP conv 0, 2 MLP conv 3, 3 MLP conv 3 chained with the projection, 4 the fused launch MLP conv 0 -> InstanceNorm -> MLP conv 3 ->
 * projection) at [batch][n] on the context's workspace; dbg == -1: gemm_f32.hip, dbg <= -2: gemm_wf.hip with its probe switches -dbg - 2 */
int imp_time_layer_gemm(imp_ctx* ctx, int batch, int n, int which, int dbg, int reps, float* ms, void* stream);
/* Pose step of the iterative loops (eval/pose_estimation.py:92-115 estimate_pose + :13-89 decompose_essential_mat) - SURVEY §8 f-1.
 * HOST arrays in, HOST arrays out (the matched keypoints of a loop iteration live on the host, eval/matching.py:68-87):
 * kpts0 / kpts1 [n][2] pixels, K0 / K1 row-major 3x3, norm_thresh in pixels (divided by the mean focal length inside).
 * `iterations` seeded minimal samples (five-point solver by default) + refits + cheirality vote on the GPU `device`, on `stream`; synchronises.
 * Returns 0 and E, R (row-major 3x3), t, *n_inliers (consensus matches in front of both cameras) and two masks [n]:
 *   mask       what the reference returns (:113-114): all True, with only the CONSENSUS entries overwritten by the cheirality result -
 *              matches outside the consensus stay 1.  The loops derive their inlier ratio and early-exit indices from it
 *              (eval/matching.py:89-90,113), so the drop-in reproduces the quirk;
 *   consensus  (optional) the geometric mask: inlier of E AND in front of both cameras.
 * 1 = no pose (fewer than 5 matches - 8 with IMP_POSE_8PT - or no consensus: the reference returns None); < 0 = error.
 * NOT OpenCV's USAC_MAGSAC: parity with that third-party solver is unpinned (see csrc/pose.hip). */
int imp_estimate_pose(const float* kpts0, const float* kpts1, int n, const double* K0, const double* K1, double norm_thresh,
                      int iterations, unsigned seed, int device, double* E, double* R, double* t, unsigned char* mask,
                      unsigned char* consensus, int* n_inliers, int flags, void* stream);
/* flags bit 0 (IMP_POSE_MAGSAC): hypotheses are ranked by the sigma-marginalised quality of MAGSAC++ (Barath et al. 2020: sum of the weights
 * w(r) of the Sampson residuals, noise scale uniform on (0, sigma_max], sigma_max = the threshold, nu = 4, k = 3.64) instead of the inlier
 * count, and the winner is refined by weighted least squares with those weights (IRLS) instead of refits on the consensus set.  The masks
 * still use the hard threshold.  This follows the PUBLISHED algorithm; OpenCV's USAC_MAGSAC implementation itself remains unpinned. */
#define IMP_POSE_MAGSAC 1
/* flags bit 1 (IMP_POSE_8PT): minimal samples of 8 matches through the linear eight-point solver (round 2) instead of 5 matches through the
 * five-point solver (Nister / Stewenius-Engels-Nister: up to 10 models per sample; csrc/pose_fivept.h), which is the default: like the
 * reference it then answers from 5 matches on (eval/pose_estimation.py:93) */
#define IMP_POSE_8PT 2
/* flags bit 2 (IMP_POSE_ADAPTIVE, round 5; five-point sampler only): adaptive termination - the rule of RANSAC / USAC that the reference reaches
 * through cv2.findEssentialMat(..., prob = 0.99999, method = cv2.USAC_MAGSAC) (eval/pose_estimation.py:96-105; OpenCV's implementation itself
 * stays unpinned).  The first 128 samples are drawn and scored, the best support gives the inlier ratio w, and only the smallest k with
 * (1 - w^5)^k <= 1e-5 samples (128 <= k <= iterations) are drawn at all - decided on the device, the call still never waits in between.
 * The samples are the same seeded sequence, so the result equals that of a fixed budget of k.  imp_pose_stats: calls / samples drawn so far. */
#define IMP_POSE_ADAPTIVE 4
void imp_pose_stats(long* calls, long* samples, int reset);
/* Chip-resident Sinkhorn health (csrc/ot_resident.hip; the kernel behind compute_score, nets/gm.py:297-303).  Its workgroups
 * exchange vectors through memory with bounded waits.  A wait that times out (a second process on the GPU, a partition mode
 * that places workgroups differently) voids the launch: the kernel poisons its outputs (maxima NaN -> mscores NaN, indices -1)
 * and raises a flag in mapped host memory.  EVERY compute entry point of the context looks at that flag first (a host read,
 * no synchronisation): if set it recovers (waits for the device, resets the exchange state, stops using the protocol that
 * failed: XCD-local launches first, then the resident kernel altogether) and returns IMP_E_RESIDENT - the results of the
 * earlier call are void, re-run the batch.
 *  imp_resident_health: the same check on demand (after the caller synchronised, e.g. after its D2H copy of the matches);
 *    *timeouts = voided launches so far, *level = 0 all protocols / 1 chip-wide exchange only / 2 streaming kernels only.
 *  imp_set_resident_verify(1) (or option ot_verify = 1): every resident launch is awaited inside the call and a voided one is
 *    re-run there on the next protocol down - calls then always return valid results, at the price of one host
 *    synchronisation per score.
 *  imp_resident_status: raw flag after a device synchronisation (tests / bench). */
int imp_resident_status(imp_ctx* ctx, int* status, int* used);
int imp_resident_health(imp_ctx* ctx, int* timeouts, int* level);
int imp_set_resident_verify(imp_ctx* ctx, int on);
/* Round 6.  With in-call recovery on (imp_set_range_recovery, the default of the Python modules) imp_match_pair / imp_match_tail[_scores]
 * also repair a waiting launch of THEIR OWN that was voided: after the call's one synchronisation the context takes its step down and the call's work is
 * enqueued again - the caller receives valid results from the same call, like every call of the reference does (nets/gm.py:145-247).
 *  imp_resident_repaired: how many calls were repaired that way.
 *  imp_resident_postmortem: the record of the LAST voided waiting launch, n <= 40 ints, returns 1 when there is one (0: none so far):
 *    [0] kind (1 Sinkhorn wait timed out, 2 Sinkhorn workgroups not spread evenly over the XCCs, 3 fused layer statistics exchange timed out)
 *    [1] tag of the launch  [2] blockIdx.x of the first waiter that gave up  [3] its HW_ID register (wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
 *    [4] its XCC  [5] phase (Sinkhorn: 1 partial column vectors, 2 half-sum swap between the two XCDs of a pair, 3 v, 4 column maxima; fused layer:
 *    1 block statistics, 2 finalised statistics)  [6] index of the granule / chunk it waited for  [7] tag it expected  [8] tag it saw
 *    [9] Sinkhorn iteration  [10] pair  [11] row group / tile  [12] groups / tiles per pair  [13] placement (0 chip-wide, 1 one XCD, 2 two XCDs per pair)
 *    [14] pairs in the launch; host side, at the moment the library noticed: [16] status word  [17] the device's spin gate was ordering several streams
 *    [18] several streams were choosing kernels  [19] consecutive sections of one stream  [20] step-down level before  [21] fused layers enabled
 *    [22] next Sinkhorn tag  [23] last fused-layer tag  [24] voided launches so far. */
/* TEST HOOK (tests/test_gpu_rehearsal.py): `workgroups` workgroups that each hold 96 KB of a CU's LDS for `microseconds` on `stream` -
 * what a collective's kernel waiting for a slower peer looks like to the launches of this library that need every CU (no reference counterpart) */
int imp_debug_hold_cus(int device, int workgroups, int microseconds, void* stream);
int imp_resident_repaired(imp_ctx* ctx);
int imp_resident_postmortem(imp_ctx* ctx, int32_t* out, int n);
/* how many calls on this context met non-finite match scores so far (reported with IMP_E_RANGE, or recovered in the call) */
int imp_range_events(imp_ctx* ctx);
/* In-call recovery from IMP_E_RANGE (round 5).  By default the library never waits for the GPU: an operand beyond the fp16 range of the
 * f16x3 arithmetic voids THAT call (all matches -1) and the NEXT entry point reports it.  With imp_set_range_recovery(ctx, 1) the one-shot
 * and tail entry points (imp_match_pair, imp_match_tail, imp_match_tail_scores) wait for their own work before they return (one host
 * synchronisation per call; not under stream capture) and a call whose scores came out non-finite is run again, inside the call, on the
 * native fp32 MFMA path (the arithmetic of imp_set_precision(ctx, 0), which has no operand limit): the caller receives what a
 * precision = f32 context computes for the same inputs.  Non-finite INPUTS stay void and are reported at the next entry point as before.
 * The Python modules switch it on by default (config key 'range_recovery'): nets/gm.py:145-247 returns finite matches for such data.
 * imp_range_recovered: how many calls were repaired that way. */
int imp_set_range_recovery(imp_ctx* ctx, int on);
int imp_range_recovered(imp_ctx* ctx);
/* for callers that compose a pass from the step API (layer calls, then score + matches per iteration) and want the same recovery: after
 * the pass, once the CALLER has synchronised its stream, imp_range_take returns 1 when a match kernel of the pass met non-finite scores,
 * clears the word (the next entry point then reports nothing) and counts the event (as recovered when `recovered` != 0: the caller is
 * about to re-run the pass under imp_set_precision(ctx, 0)); 0 otherwise.  recovered == 2: only counts a recovery (for a pass whose event one of
 * its own later entry points already reported as IMP_E_RANGE).  The Python modules do this for the all-iterations path. */
int imp_range_take(imp_ctx* ctx, int recovered);
/* how many times the tag counter of hipGraph-REPLAYED resident launches wrapped (about every 7 million replays at 100 Sinkhorn iterations;
 * the library clears the exchange buffers at the next entry point - not an error).  Test hook: option ot_graph_tag0=<first tag>. */
int imp_tag_wraps(imp_ctx* ctx);

/* ---------------------------------------------------------------------------------------------------------------------------
 * SuperPoint front-end: nets/superpoint.py:97-232.  Its own handle (independent of imp_ctx);
 * bound to one device, not thread-safe, one image batch in flight.  Keypoint counts are data dependent, so the call is split
 * the way the reference's variable-length lists force it to be: imp_sp_detect runs the network, NMS, threshold / border filter
 * and top-k and returns the per-image counts (synchronises `stream`); the caller allocates outputs of those sizes and
 * imp_sp_describe fills them (asynchronous on `stream`). */
typedef struct imp_sp_ctx imp_sp_ctx;
/* nets/superpoint.py:112-137 (module construction): descriptor_dim = config['descriptor_dim'] (64 / 128 / 192 / 256) */
int imp_sp_create(imp_sp_ctx** out, int device, int descriptor_dim);
void imp_sp_destroy(imp_sp_ctx* ctx);
/* nets/superpoint.py:155-156 (load_state_dict): name = state_dict key ("conv1a.weight" ... "convDb.bias"), HOST float32 data in
 * PyTorch layout ([out][in][k][k] for weights); imp_sp_finalize packs them for the kernels (split-half MFMA B fragments) */
int imp_sp_set_weight(imp_sp_ctx* ctx, const char* name, const float* data, int64_t count);
int imp_sp_finalize(imp_sp_ctx* ctx);
/* nets/superpoint.py:170-217 (forward up to top_k_keypoints): image = DEVICE float32 [B][1][H][W] (H, W >= 8; sizes that are not
 * multiples of 8 floor exactly as the three MaxPool2d(2, 2) do), nms_radius / keypoint_threshold / max_keypoints (-1 = all, at
 * most 16384) / remove_borders = the config keys of :104-110; align_corners = what the caller resolved for :89 (the reference
 * decides from the installed torch version string).  counts = HOST int[B].  Synchronises. */
int imp_sp_detect(imp_sp_ctx* ctx, const float* image, int B, int H, int W, int nms_radius, float keypoint_threshold,
                  int max_keypoints, int remove_borders, int align_corners, void* stream, int* counts);
/* nets/superpoint.py:219-232 for image b of the last imp_sp_detect: DEVICE outputs keypoints [n][2] (x, y), scores [n],
 * descriptors [D][n] (the reference's layout), n = counts[b]; n == 0 writes nothing */
int imp_sp_describe(imp_sp_ctx* ctx, int b, float* keypoints, float* scores, float* descriptors, void* stream);
/* nets/superpoint.py:140-168 (extract) and test probes, from the last imp_sp_detect; each output optional (NULL):
 * scores [B][8h][8w] dense (before NMS), nms_scores [B][8h][8w] (after simple_nms), descriptors [B][D][h][w] L2-normalised */
int imp_sp_dense(imp_sp_ctx* ctx, float* scores, float* nms_scores, float* descriptors, void* stream);
/* test entry: ONE convolution of the stack on caller data, NHWC float32 device tensors: layer 0 = conv1a (in [B][H][W], out
 * [B][H][W][64], always ReLU), 1..7 = conv1b, conv2a, conv2b, conv3a, conv3b, conv4a, conv4b, 8 = convPa|convDa stacked (512
 * output channels), 9 = convDb (1x1).  in [B][H][W][Cin] -> out [B][Ho][Wo][Cout], pool != 0 fuses MaxPool2d(2, 2) (Ho = H / 2). */
int imp_sp_op_conv(imp_sp_ctx* ctx, int layer, const float* in, int B, int H, int W, float* out, int relu, int pool, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMP_HIP_H */
