// libimp_hip.so: context, weight packing and kernel orchestration behind the C-ABI of include/imp_hip.h.
// Host-side C++ only (no torch); every launch goes to the caller's stream and nothing here synchronises
// the host except imp_create / imp_finalize_weights / imp_reserve (allocation + upload) and imp_time_*.
#include "../../include/imp_hip.h"
#include "imp_kernels.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(IMP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };
struct Linear { float* W = nullptr; float* b = nullptr; int out = 0, in = 0; };
struct NormC { float* mean = nullptr; float* rstd = nullptr; float* gamma = nullptr; float* beta = nullptr; int c = 0; };
struct GnnLayer {
    bool cross = false, shared = false;
    Linear proj;     // non-shared: packed [3D][D] (q|k|v, head-major rows); shared: value projection [D][D]
    Linear merge;    // [D][D], input columns head-major
    Linear mlp0;     // [2D][2D]
    Linear mlp0f;    // [2D][2D] with the merge conv folded in: [W1a | W1b.Wm], bias b1 + W1b.bm (fuse_merge)
    NormC bn;        // only norm_fn == 'bn'
    Linear mlp3;     // [D][2D]
    // the same weights as split-half MFMA fragments (gemm_wf.hip: wf_pack)
    _Float16 *proj_wf = nullptr, *mlp0f_wf = nullptr, *mlp3_wf = nullptr;
};
struct AttnCache {   // cached operands of the last non-shared layer of a kind (self / cross)
    bool valid = false;
    int batch = 0, n[2] = {0, 0};
    bool masked[2] = {false, false};   // key mask of image 0 / image 1 in effect when it was computed
    bool kv_image = false;             // the k | v slots of qkv[kind] hold split-half images (AttnParams::kv_planes), not fp32
};

}  // namespace

struct imp_ctx {
    imp_config cfg{};
    int device = 0, D = 0, dh = 0;
    std::vector<std::string> schema;
    std::map<std::string, HostTensor> raw;
    bool finalized = false;
    int prec = 1;             // matrix arithmetic: 1 = f16x3 split (default), 0 = native fp32 MFMA (imp_set_precision / IMP_PRECISION=f32)
    int ot_compact = 0;       // Sinkhorn iterations stream the 3-byte copy of P (imp_set_sinkhorn_storage)
    bool fuse_merge = true;   // fold attn.merge into mlp.0 (one GEMM and one launch less per layer) (always on since round 4)
    float bin_score = 1.f;
    std::vector<void*> allocs_w, allocs_ws, allocs_x;   // weights / workspace (regrown) / resident-Sinkhorn exchange
    // packed weights
    std::vector<Linear> kenc;
    std::vector<NormC> kenc_bn;
    std::vector<GnnLayer> layers;
    std::vector<Linear> final_proj;
    // workspace (capacity: cap_b pairs x cap_n keypoints per image)
    int cap_b = 0, cap_n = 0, kenc_maxc = 0;
    float *qkv[2][2] = {}, *lse[2][2] = {};         // [kind: 0 self, 1 cross][side]
    uint8_t* cmask[2][2] = {};                       // [kind][image] cached key masks
    float *attn_out[2] = {}, *msg[2] = {}, *hid[2] = {}, *stats[2] = {}, *nstat[2] = {};
    float *kbuf[2][2] = {};                          // keypoint-encoder ping-pong per side
    float *descw[2] = {}, *mdesc[2] = {}, *nkp[2] = {};
    float* dist = nullptr;
    OtBuffers ot{};
    // chip-resident Sinkhorn (ot_resident.hip): exchange buffers sized for one workgroup per CU, independent of N
    int ot_resident = 1;      // IMP_OT_RESIDENT=0 forces the streaming path
    int num_cus = 0;
    float *xpart = nullptr, *xv = nullptr, *xmax = nullptr;
    unsigned xtag = 0;       // tag base of the next resident launch (tags must never repeat on the exchange buffers)
    int ot_graph = 1;
    int* xstatus = nullptr;                        // device, 32 words: [0] time-out flag, [4..11] per-XCC ticket counters of the LOCAL launches;
                                                   //   launches recorded into a hipGraph: [16] tag base, [17] ticket base, [18] workgroups done, [20..27] their tickets
    int* xstatus_host = nullptr;                   // the same flag in mapped host memory: read at every entry without synchronising
    int* xstatus_hostdev = nullptr;                //   its device address
    int* range_host = nullptr;                     // word 1 of the same page: a match kernel saw non-finite scores (IMP_E_RANGE)
    int* range_hostdev = nullptr;
    unsigned graph_tag0 = 0x80000000u;             // first tag of the graph launches (option ot_graph_tag0: a test starts close to the wrap)
    int tag_wraps = 0;                             // word 2 of the page: the graph launches' tag counter wrapped (resident_health clears the buffers)
    unsigned ticket_base = 0;                      // value of the per-XCC ticket counters before the next LOCAL launch
    int num_xccs = 0;
    int ot_degrade = 0;      // raised by a time-out: 1 = no XCD-local launches any more (chip-wide exchange only), 2 = streaming kernels only
    int ot_verify = 0;       // option ot_verify = 1 / imp_set_resident_verify: wait for every resident launch and re-run a voided one inside the call
    int ot_fake = 0;         // TEST HOOK option ot_fake_placement = 1: LOCAL workgroups lie about their XCC (forces the time-out path)
    int resident_timeouts = 0;
    int resident_repaired = 0;   // voided waiting launches whose call was run again INSIDE the call (range_recover_in_call)
    int postmortem[40] = {};     // the last voided launch's record: [0..14] the kernel's (imp_kernels.h imp_postmortem_write), [16..] what the host knew when it noticed
    int postmortem_valid = 0;
    int range_events = 0;
    int range_recover = 0;                         // imp_set_range_recovery: the one-shot / tail entry points wait for their own work and re-run a call whose operands left the fp16 range on the fp32 MFMA path
    int range_recovered = 0;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int xcap_b = 0;
    float *max0 = nullptr, *max1 = nullptr, *colpart_v = nullptr;
    int *arg0 = nullptr, *arg1 = nullptr, *colpart_i = nullptr;
    float *colsum[4] = {}, *amass[4] = {}, *mass[2] = {};
    AttnCache cache[2];
    float* xhalf = nullptr;  // resident Sinkhorn, two XCDs per pair: the half sums the XCDs swap ([2 iteration parities][8 vectors]: ot_resident.hip, phase C)
    int ot_hier = 1;         // two-XCDs-per-pair resident launches (hierarchical column sums) when a pair fits 64 CUs and B <= 4 (option ot_hier = 0 disables)
    int ot_local = 1;        // XCD-local resident Sinkhorn launches when a pair fits one XCD (option ot_local = 0 disables)
    unsigned* stat_cnt = nullptr;   // [cap_b][2][WF_MAX_PSPLIT] tickets of the statistics merge inside the MLP0 launch (gemm_wf.hip)
    int kv_image = 1;        // option kv_image = 0: projections write K / V as fp32 (round-2 format) instead of the split-half image attention copies
    float* kf32[2] = {};     // per image: K of a cached attention converted back to fp32 [B][n][D] (pooling / probability readers)
    int wf_chain = 1;        // option wf_chain = 0: never compute the next layer's projection inside the MLP3 launch
    long wf_chain_min_tiles = 160, wf_max_tiles = 640, wf_proj_max_tiles = 40;     // (option wf_chain_min; the other two are fixed)
    int wf_fused = 1;        // IMP_WF_FUSED=0: never run a layer's MLP0 -> InstanceNorm -> MLP3 (-> next projection) as ONE launch (gemm_wf.hip fused kernel)
    long wf_fused_min_tiles = 100;   // option wf_fused_min (measured: the fused launch wins from ~64 tiles up - 128 tiles: -7..-9 %, 64: +-1 % - and loses 12-18 % at 16-32)
    int probe_prof = 0;      // option probe_prof
    int attn_share_mode = 0; // option attn_shares (AttnParams::share_mode)
    int wf_fused_fake = 0;   // TEST HOOK (option wf_fused_fake): one workgroup of every fused launch withholds its statistics (forces the time-out path)
    float *fx_rec[2] = {}, *fx_fin[2] = {};   // fused layer: statistics granules [B][tiles][512] x 16 B and (mean, rstd) granules [B][512] x 16 B per image
    unsigned fx_tag = 0;     // tag of the last fused launch (tags never repeat on fx_rec / fx_fin)
    size_t fx_rec_floats = 0, fx_fin_floats = 0;
    int* fx_status = nullptr;   // device word the waiters of a fused launch watch (3 = a wait timed out)
    // native lock-step loop (imp_loop_lockstep): device outputs of a scored iteration, their pinned mirror, the pose workers
    int64_t* lp_idx = nullptr; float* lp_ms = nullptr; unsigned char* lp_pin = nullptr; size_t lp_cap = 0;
    hipEvent_t lp_ev = nullptr;
    struct PoseWorkers* lp_workers = nullptr;
    // native EIMP loop (imp_loop_lockstep_uncertainty): score tensors of a scored iteration, the pool's id lists + counts (device and pinned
    // mirror), the second descriptor buffers the kept rows are gathered into
    float* lu_scores = nullptr; size_t lu_scores_cap = 0;
    int64_t* lu_pool = nullptr; int64_t* lu_pool_pin = nullptr; size_t lu_pool_cap = 0;
    float* lu_alt[2] = {nullptr, nullptr}; size_t lu_alt_cap[2] = {0, 0};
    RaggedCounts rc{};       // imp_set_counts: per-pair keypoint counts of the NEXT calls (rc.on = 0: uniform batches); rc_batch pairs
    int rc_batch = 0;
    int ot_lane = 0;         // (rounds 2-3, kept for the graph-free fallback, never set:) resident Sinkhorn launches go through the device's lane stream (rounds 2-3) instead of the caller's stream under the spin gate
    int use_wf = 1;          // weight-fragment GEMMs (gemm_wf.hip) for the layer convolutions when f16x3, D = 256, relu + InstanceNorm; option gemm_wf = 0 disables
    float* attn_split_ws = nullptr;        // key-split scratch of the attention kernel (grown on demand, allocs_x)
    unsigned* attn_split_cnt = nullptr;
    size_t attn_split_cap = 0, attn_split_units = 0;
};

namespace {

template <typename T>
int dev_alloc(imp_ctx* c, std::vector<void*>& pool, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T) + 64);
    if (e != hipSuccess) return fail(IMP_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    // debugging aid (IMP_POISON_WORKSPACE=1): fresh allocations are filled with NaN bit patterns, so that a read of
    // never-written workspace shows up in the results instead of depending on what the allocator handed out
    static const bool poison = [] { const char* v = getenv("IMP_POISON_WORKSPACE"); return v && atoi(v) != 0; }();
    if (poison) { (void)hipMemset(p, 0xFF, count * sizeof(T) + 64); (void)hipDeviceSynchronize(); }
    pool.push_back(p);
    *out = static_cast<T*>(p);
    return IMP_OK;
}
void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

std::string kname(const char* fmt, int a, int b = 0) {
    char buf[128];
    snprintf(buf, sizeof buf, fmt, a, b);
    return buf;
}
bool layer_shared(const imp_config& cfg, int li) {
    if (cfg.model == IMP_MODEL_GM) return false;
    if (li < 4) return false;                 // [F,F]*2 + [F,F,T,T]*21   nets/gms.py:17, nets/adgm.py:18
    return ((li - 4) & 3) >= 2;
}
int n_kenc(const imp_config& cfg) {
    int n = 0;
    while (n < 8 && cfg.kenc_channels[n] > 0) ++n;
    return n;
}

void build_schema(imp_ctx* c) {
    const imp_config& cfg = c->cfg;
    std::vector<std::string>& s = c->schema;
    s.clear();
    s.push_back("bin_score");
    const int nk = n_kenc(cfg);
    for (int i = 0; i <= nk; ++i) {
        s.push_back(kname("kenc.encoder.%d.weight", 3 * i));
        s.push_back(kname("kenc.encoder.%d.bias", 3 * i));
        if (i < nk && cfg.norm_fn == IMP_NORM_BN)
            for (const char* f : {"weight", "bias", "running_mean", "running_var"})
                s.push_back(kname("kenc.encoder.%d.", 3 * i + 1) + f);
    }
    for (int li = 0; li < cfg.n_gnn_layers; ++li) {
        const std::string p = kname("gnn.layers.%d", li);
        if (layer_shared(cfg, li)) {
            for (const char* f : {".proj.weight", ".proj.bias", ".merge.weight", ".merge.bias"}) s.push_back(p + f);
        } else {
            for (const char* f : {".attn.merge.weight", ".attn.merge.bias"}) s.push_back(p + f);
            for (int j = 0; j < 3; ++j) {
                s.push_back(p + kname(".attn.proj.%d.weight", j));
                s.push_back(p + kname(".attn.proj.%d.bias", j));
            }
        }
        for (const char* f : {".mlp.0.weight", ".mlp.0.bias", ".mlp.3.weight", ".mlp.3.bias"}) s.push_back(p + f);
        if (cfg.norm_fn == IMP_NORM_BN)
            for (const char* f : {".mlp.1.weight", ".mlp.1.bias", ".mlp.1.running_mean", ".mlp.1.running_var"})
                s.push_back(p + f);
    }
    for (int i = 0; i < cfg.n_layers; ++i) {
        s.push_back(kname("final_proj.%d.weight", i));
        s.push_back(kname("final_proj.%d.bias", i));
    }
}

int upload(imp_ctx* c, float** dst, const std::vector<float>& host) {
    int rc = dev_alloc(c, c->allocs_w, dst, host.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*dst, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return IMP_OK;
}
const HostTensor* get(imp_ctx* c, const std::string& key, int64_t numel) {
    auto it = c->raw.find(key);
    if (it == c->raw.end()) { g_err = "missing state_dict key: " + key; return nullptr; }
    if ((int64_t)it->second.data.size() != numel) {
        g_err = "state_dict tensor " + key + " has " + std::to_string(it->second.data.size()) + " elements, expected " +
                std::to_string(numel);
        return nullptr;
    }
    return &it->second;
}
int upload_wf(imp_ctx* c, _Float16** dst, const float* W, int N, int K) {
    if (!gemm_wf_supported(K, N)) { *dst = nullptr; return IMP_OK; }
    std::vector<_Float16> h((size_t)N * K * 2);
    wf_pack(W, N, K, h.data());
    int rc = dev_alloc(c, c->allocs_w, dst, h.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*dst, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    return IMP_OK;
}
// plain [out][in] linear
int pack_linear(imp_ctx* c, const std::string& prefix, int out, int in, Linear* L) {
    const HostTensor* w = get(c, prefix + ".weight", (int64_t)out * in);
    const HostTensor* b = get(c, prefix + ".bias", out);
    if (!w || !b) return IMP_E_KEY;
    L->out = out; L->in = in;
    int rc = upload(c, &L->W, w->data);
    if (rc) return rc;
    return upload(c, &L->b, b->data);
}
// head permutation: reference channel c = d*H + h  ->  packed channel h*dh + d   (nets/layers.py:119-120)
inline int ref_channel(int packed, int dh) { return (packed % dh) * IMP_NUM_HEADS + packed / dh; }

int pack_norm(imp_ctx* c, const std::string& prefix, int ch, NormC* N) {
    const HostTensor* g = get(c, prefix + ".weight", ch);
    const HostTensor* b = get(c, prefix + ".bias", ch);
    const HostTensor* rm = get(c, prefix + ".running_mean", ch);
    const HostTensor* rv = get(c, prefix + ".running_var", ch);
    if (!g || !b || !rm || !rv) return IMP_E_KEY;
    std::vector<float> rstd(ch);
    for (int i = 0; i < ch; ++i) rstd[i] = 1.0f / std::sqrt(rv->data[i] + 1e-3f);   // eps = 1e-3 nets/layers.py:70
    N->c = ch;
    int rc;
    if ((rc = upload(c, &N->mean, rm->data))) return rc;
    if ((rc = upload(c, &N->rstd, rstd))) return rc;
    if ((rc = upload(c, &N->gamma, g->data))) return rc;
    return upload(c, &N->beta, b->data);
}

int ensure_workspace(imp_ctx* c, int batch, int n) {
    if (batch <= c->cap_b && n <= c->cap_n) return IMP_OK;
    if (batch < c->cap_b) batch = c->cap_b;
    if (n < c->cap_n) n = c->cap_n;
    HIP_TRY(hipDeviceSynchronize());
    free_pool(c->allocs_ws);
    c->cap_b = c->cap_n = 0;
    c->cache[0].valid = c->cache[1].valid = false;
    const size_t B = batch, N = n, D = c->D;
    const size_t tiles = (N + 31) / 32;            // statistics blocks: one per 32 rows at the smallest GEMM tile
    int rc = 0;
    for (int k = 0; k < 2 && !rc; ++k)
        for (int s = 0; s < 2 && !rc; ++s) {
            rc = dev_alloc(c, c->allocs_ws, &c->qkv[k][s], B * N * 3 * D);
            if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->lse[k][s], B * IMP_NUM_HEADS * N);
            if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->cmask[k][s], B * N);
        }
    for (int s = 0; s < 2 && !rc; ++s) {
        rc = dev_alloc(c, c->allocs_ws, &c->attn_out[s], B * N * D);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->msg[s], B * N * D);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->hid[s], B * N * 2 * D);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->stats[s], B * tiles * 2 * D * 2 + 2 * B * tiles * (size_t)c->kenc_maxc * 2);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->nstat[s], B * (size_t)(2 * D > (size_t)c->kenc_maxc ? 2 * D : c->kenc_maxc) * 2);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->kbuf[s][0], B * N * (size_t)c->kenc_maxc);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->kbuf[s][1], B * N * (size_t)c->kenc_maxc);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->descw[s], B * N * D);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->mdesc[s], B * N * D);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->nkp[s], B * N * 2);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->mass[s], N + 4);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->kf32[s], B * N * D);
    }
    const size_t ld = (N + 1 + 3) & ~(size_t)3;
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->dist, B * N * N);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.P, B * (N + 1) * ld);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.PT, B * (N + 1) * ld);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.u, B * ld);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.v, B * ld);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.v2, B * ld);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.partials, B * ((N + 1 + 15) / 16) * ld);   // >= ceil(n0 / FP_ROWS) partial vectors
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->ot.P24, B * (N + 1) * (ld / 4) * 3);          // 3 bytes per matrix element
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->max0, B * N);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->max1, B * N);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->arg0, B * N);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->arg1, B * N);
    const size_t chunks = score_maxima_chunks((int)N);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->colpart_v, B * chunks * N);
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->colpart_i, B * chunks * N);
    for (int k = 0; k < 4 && !rc; ++k) {
        rc = dev_alloc(c, c->allocs_ws, &c->colsum[k], B * IMP_NUM_HEADS * N);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->amass[k], B * N);
    }
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->stat_cnt, (B * 2 + 2) * WF_MAX_PSPLIT);
    // fused layer MLP (gemm_wf.hip gemm_wf_fused_kernel): granule buffers of the statistics exchange, tag 0 = never written
    c->fx_rec_floats = B * ((N + 63) / 64) * 2 * D * 4;
    c->fx_fin_floats = B * 2 * D * 4;
    for (int s = 0; s < 2 && !rc; ++s) {
        rc = dev_alloc(c, c->allocs_ws, &c->fx_rec[s], c->fx_rec_floats);
        if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->fx_fin[s], c->fx_fin_floats);
        if (!rc) {
            HIP_TRY(hipMemset(c->fx_rec[s], 0, c->fx_rec_floats * sizeof(float)));
            HIP_TRY(hipMemset(c->fx_fin[s], 0, c->fx_fin_floats * sizeof(float)));
        }
    }
    if (!rc) rc = dev_alloc(c, c->allocs_ws, &c->fx_status, 16);
    if (!rc) HIP_TRY(hipMemset(c->fx_status, 0, 64));
    if (!rc) {
        HIP_TRY(hipMemset(c->stat_cnt, 0, (B * 2 + 2) * WF_MAX_PSPLIT * sizeof(unsigned)));
        HIP_TRY(hipDeviceSynchronize());              // the clear runs on the NULL stream, which the callers' streams do not wait for
    }
    if (rc) { free_pool(c->allocs_ws); return rc; }
    c->cap_b = batch;
    c->cap_n = n;
    return IMP_OK;
}

inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }
int resident_health(imp_ctx* c);

int check_ready(imp_ctx* c, int batch, int n0, int n1) {
    if (!c) return fail(IMP_E_ARG, "null context");
    if (!c->finalized) return fail(IMP_E_STATE, "weights not finalised (imp_finalize_weights)");
    if (batch < 1 || n0 < 1 || n1 < 1) return fail(IMP_E_ARG, "batch, n0, n1 must be >= 1");
    HIP_TRY(hipSetDevice(c->device));
    if (int hrc = resident_health(c)) return hrc;
    if (c->rc.on) {
        if (batch != c->rc_batch) return fail(IMP_E_ARG, "per-pair keypoint counts are set for " + std::to_string(c->rc_batch) + " pairs (imp_set_counts), the call has " + std::to_string(batch));
        for (int b = 0; b < batch; ++b)
            if (c->rc.n[0][b] > n0 || c->rc.n[1][b] > n1) return fail(IMP_E_ARG, "a per-pair keypoint count (imp_set_counts) exceeds the padded size of the call");
    }
    return ensure_workspace(c, batch, n0 > n1 ? n0 : n1);
}
// the per-pair counts of a ragged batch into a launch's parameter block (side s = image s unless the launch says otherwise)
template <typename P> void apply_ragged(const imp_ctx* c, P& p) {
    p.rc = c->rc;
    p.side[0].img = 0; p.side[1].img = 1;
}

// attention launch; in f16x3 mode the context lends its key-split scratch (grown on demand) for launches too small to fill the chip
int launch_attention(imp_ctx* c, AttnParams& a, int batch, hipStream_t st) {
    if (c->prec != 1) { HIP_TRY(launch_attention_f32(a, batch, st)); return IMP_OK; }
    a.share_mode = c->attn_share_mode;
    // size the scratch for the largest split the launcher may choose (it decides with the pointers set)
    float dummy_ws; unsigned dummy_cnt;
    a.split_ws = &dummy_ws; a.split_cnt = &dummy_cnt;
    const int ns = attention_f16x3_splits(a, batch);
    a.split_ws = nullptr; a.split_cnt = nullptr;
    if (ns > 1) {
        const size_t need = attention_f16x3_split_floats(a, batch, ns), units = attention_f16x3_split_units(a, batch);
        if (need > c->attn_split_cap || units > c->attn_split_units) {
            HIP_TRY(hipStreamSynchronize(st));                 // (rare: the scratch grows; earlier launches may still use the old one)
            int rc = dev_alloc(c, c->allocs_x, &c->attn_split_ws, need);
            if (!rc) rc = dev_alloc(c, c->allocs_x, &c->attn_split_cnt, units);
            if (rc) return rc;
            HIP_TRY(hipMemset(c->attn_split_cnt, 0, units * sizeof(unsigned)));
            HIP_TRY(hipDeviceSynchronize());                   // the clear runs on the NULL stream
            c->attn_split_cap = need; c->attn_split_units = units;
        }
        a.split_ws = c->attn_split_ws; a.split_cnt = c->attn_split_cnt;
    }
    HIP_TRY(launch_attention_f16x3(a, batch, st));
    return IMP_OK;
}
WfParams wf_defaults() {
    WfParams p;
    memset(&p, 0, sizeof p);
    p.kv_image_col = p.kv_image_col2 = 1 << 30;         // no column is written as a split-half image
    return p;
}
GemmParams gemm_defaults(const imp_ctx* c, int K) {
    GemmParams p;
    memset(&p, 0, sizeof p);
    p.prec = c->prec;
    p.K = K; p.ksplit = K; p.nside = 2; p.nsub = 1; p.div = 1.f; p.norm_eps = 1e-3f;
    return p;
}

// y_s = x_s @ W^T + b for both image sides (the plain 1x1 conv)
int linear_both(imp_ctx* c, const Linear& L, int batch, const int n[2], const float* const x[2], int ldx, float* const y[2],
                int ldy, hipStream_t st) {
    GemmParams p = gemm_defaults(c, L.in);
    for (int s = 0; s < 2; ++s) {
        GemmSide& g = p.side[s];
        g.A = x[s]; g.W = L.W; g.C = y[s]; g.M = n[s]; g.N = L.out;
        g.sA_b = (long)n[s] * ldx; g.sC_b = (long)n[s] * ldy;
    }
    p.bias = L.b; p.lda = ldx; p.ldw = L.in; p.ldc = ldy;
    apply_ragged(c, p);
    HIP_TRY(launch_gemm_f32(p, batch, st));
    return IMP_OK;
}

// keypoint encoder: first conv on the VALU, then the MLP chain as GEMMs with InstanceNorm/BatchNorm + activation
// applied while staging the A operand and the statistics produced by the previous GEMM's epilogue.
int run_kenc(imp_ctx* c, int batch, const int n[2], const float* const kpts[2], const float* const scores[2],
             float width, float height, const float* const desc[2], float* const out[2], hipStream_t st) {
    const imp_config& cfg = c->cfg;
    const int nk = n_kenc(cfg);
    const int D = c->D;
    const bool in_norm = cfg.norm_fn == IMP_NORM_IN;
    // two statistics slots per side behind the GNN-layer statistics: a launch reads the previous layer's slot in its
    // prologue while other workgroups already write this layer's slot in their epilogue
    const size_t tiles_cap = ((size_t)c->cap_n + 31) / 32;
    const size_t slot = (size_t)c->cap_b * tiles_cap * c->kenc_maxc * 2;
    float* kstats[2] = {c->stats[0] + (size_t)c->cap_b * tiles_cap * 2 * D * 2,
                        c->stats[1] + (size_t)c->cap_b * tiles_cap * 2 * D * 2};
    Kenc0Side ks[2];
    for (int s = 0; s < 2; ++s) ks[s] = Kenc0Side{kpts[s], scores[s], c->kbuf[s][0], in_norm ? kstats[s] : nullptr, n[s]};
    HIP_TRY(launch_kenc_first(ks, batch, c->kenc[0].out, c->kenc[0].W, c->kenc[0].b, width, height, st, &c->rc));
    int in_rows = 64;                                            // kenc_first: 64-token statistics blocks
    const int maxn = n[0] > n[1] ? n[0] : n[1];
    int cur = 0;
    for (int i = 1; i <= nk; ++i) {
        const Linear& L = c->kenc[i];
        const bool last = i == nk;
        GemmParams p = gemm_defaults(c, L.in);
        p.flags = GEMM_PRO_NORM;
        p.act = cfg.ac_fn;
        if (!in_norm) {
            const NormC& nc = c->kenc_bn[i - 1];
            p.flags |= GEMM_PRO_AFFINE;
            p.nm_mean = nc.mean; p.nm_rstd = nc.rstd; p.nm_gamma = nc.gamma; p.nm_beta = nc.beta;
        }
        if (!last && in_norm) p.flags |= GEMM_EPI_STATS;
        const int srows = gemm_stats_rows(maxn, L.out, 2 * batch);
        if (in_norm) {
            StatsSide ss[2];
            for (int s = 0; s < 2; ++s) ss[s] = StatsSide{kstats[s] + ((i - 1) & 1) * slot, c->nstat[s], (n[s] + in_rows - 1) / in_rows, n[s], in_rows};
            HIP_TRY(launch_stats_finalize(ss, 2, batch, L.in, 1e-3f, st, &c->rc));
        }
        for (int s = 0; s < 2; ++s) {
            GemmSide& g = p.side[s];
            g.A = c->kbuf[s][cur]; g.W = L.W; g.M = n[s]; g.N = L.out;
            g.sA_b = (long)n[s] * L.in;
            g.in_stats = in_norm ? c->nstat[s] : nullptr;
            if (last) {
                g.C = out[s]; g.sC_b = (long)n[s] * D;
                if (desc[s]) { g.R = desc[s]; g.sR_b = (long)n[s] * D; }
            } else {
                g.C = c->kbuf[s][cur ^ 1]; g.sC_b = (long)n[s] * L.out;
                g.out_stats = in_norm ? kstats[s] + (i & 1) * slot : nullptr;
            }
        }
        p.bias = L.b; p.lda = L.in; p.ldw = L.in; p.ldc = last ? D : L.out; p.ldr = D;
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_f32(p, batch, st));
        in_rows = srows;
        cur ^= 1;
    }
    return IMP_OK;
}

// column-pass split of a weight-fragment GEMM launch: enough workgroups to cover the chip (the largest divisor of the pass count
// that keeps tiles x split at or under ~1.5 workgroups per CU)
int wf_pass_split(const imp_ctx* c, long tiles, int N) {
    const int npass = N / 128;
    int best = 1;
    for (int d = 1; d <= npass; ++d)
        if (npass % d == 0 && tiles * d <= (long)c->num_cus * 3 / 2) best = d;
    return best;
}

struct SpinGate;
SpinGate* spin_gate(int device);
bool spin_gate_shared(int device, hipStream_t st);
int spin_enter(SpinGate* g, hipStream_t st);
int spin_leave(SpinGate* g, hipStream_t st);

// tag of the next fused layer launch: unique on the context's granule buffers (cleared before the 32-bit count wraps)
unsigned next_fused_tag(imp_ctx* c) {
    if (c->fx_tag >= 0xFFFFFF00u) {
        (void)hipDeviceSynchronize();
        for (int s = 0; s < 2; ++s) {
            (void)hipMemset(c->fx_rec[s], 0, c->fx_rec_floats * sizeof(float));
            (void)hipMemset(c->fx_fin[s], 0, c->fx_fin_floats * sizeof(float));
        }
        (void)hipDeviceSynchronize();
        c->fx_tag = 0;
    }
    return ++c->fx_tag;
}

// steps 4 + 5 of a layer (and the next layer's projection) as ONE launch (gemm_wf.hip gemm_wf_fused_kernel), inside a gate section of the
// device: mlp.0 on cat[x, attention output] -> InstanceNorm statistics exchanged between the tiles of an image -> ReLU -> mlp.3 + bias +
// residual -> `out` (-> q|k|v of layer NL into nqkv).  NL == nullptr: no chained projection
int launch_fused_layer(imp_ctx* c, const GnnLayer& L, int batch, const int n[2], const float* const desc[2], float* const out[2],
                       const GnnLayer* NL, float* const* nqkv, bool next_image, hipStream_t st, unsigned long long* prof = nullptr) {
    const int D = c->D;
    WfParams p;
    memset(&p, 0, sizeof p);
    p.kv_image_col = p.kv_image_col2 = 1 << 30;
    p.K = 2 * D; p.ksplit = D; p.N = 2 * D; p.nside = 2;
    for (int s = 0; s < 2; ++s) {
        WfSide& g = p.side[s];
        g.A = desc[s]; g.A2 = c->attn_out[s]; g.C = out[s]; g.R = desc[s]; g.M = n[s];
        g.sA_b = (long)n[s] * D; g.sA2_b = (long)n[s] * D; g.sC_b = (long)n[s] * D; g.sR_b = (long)n[s] * D;
        if (NL) { g.C2 = NL->shared ? nqkv[s] + 2 * D : nqkv[s]; g.sC2_b = (long)n[s] * 3 * D; }
    }
    p.norm_eps = 1e-3f;
    p.Wf_ = L.mlp0f_wf; p.bias = L.mlp0f.b; p.lda = D; p.lda2 = D; p.ldc = D; p.ldr = D;
    if (NL) {
        p.Wf2_ = NL->proj_wf; p.bias2 = NL->proj.b; p.N2 = NL->proj.out; p.ldc2 = 3 * D;
        if (next_image) p.kv_image_col2 = NL->shared ? 0 : D;
    }
    p.pass_split = 1;
    apply_ragged(c, p);
    WfFused f;
    memset(&f, 0, sizeof f);
    f.Wf3_ = L.mlp3_wf; f.bias3 = L.mlp3.b;
    for (int s = 0; s < 2; ++s) { f.rec[s] = c->fx_rec[s]; f.fin[s] = c->fx_fin[s]; }
    f.tag = next_fused_tag(c);
    f.status = c->fx_status; f.host_status = c->xstatus_hostdev;
    f.fake = c->wf_fused_fake;
    f.prof = prof;
    SpinGate* gate = spin_gate(c->device);
    if (!gate) return fail(IMP_E_HIP, "spin gate: cannot create events");
    if (int grc = spin_enter(gate, st)) return grc;
    const hipError_t e = launch_gemm_wf_fused(p, f, batch, st);
    const int lrc = spin_leave(gate, st);
    HIP_TRY(e);
    return lrc;
}

// One GNN layer (nets/layers.py:139-149 / :182-218) on both images.
//   proj_done: this layer's q|k|v (or value) projection was already produced by the previous layer's chained launch
//   chain_li:  >= 0: the caller will run layer chain_li next ON THE OUTPUT OF THIS CALL, unmodified and with no pooling in between:
//              its projection may be computed here, in the epilogue tile of this layer's last convolution (gemm_wf.hip CHAIN);
//              *chained reports whether it was
int run_layer(imp_ctx* c, int li, int batch, const int n[2], const float* const desc[2], float* const out[2],
              const uint8_t* const kmask[2], hipStream_t st, bool proj_done = false, int chain_li = -1, bool* chained = nullptr) {
    const imp_config& cfg = c->cfg;
    const GnnLayer& L = c->layers[li];
    const int D = c->D, kind = L.cross ? 1 : 0;
    AttnCache& cache = c->cache[kind];
    float* const* qkv = c->qkv[kind];
    if (chained) *chained = false;
    if (L.shared) {
        if (!cache.valid || cache.batch != batch || cache.n[0] != n[0] || cache.n[1] != n[1])
            return fail(IMP_E_STATE, "attention-sharing layer " + std::to_string(li) +
                                         " called without a matching cached attention of the same kind/shape");
    }
    // weight-fragment GEMMs: the default for the f16x3 arithmetic (gemm_wf.hip)
    // (64-row tiles with the whole K in LDS, 8 waves in two groups that alternate between K loop and epilogue; small launches deal
    // the column passes of a tile to several workgroups, wf_pass_split.  option gemm_wf: 0 never, 1 by this rule (default), 2 always)
    const long wf_tiles = (long)batch * ((n[0] + 63) / 64 + (n[1] + 63) / 64);
    const bool wf = c->prec == 1 && c->use_wf && (c->use_wf > 1 || wf_tiles <= c->wf_max_tiles);
    // K / V as split-half images (round 3): the projection's epilogue writes, per 64-channel head segment, [64 hi halves | 64 lo halves]
    // - exactly what every attention workgroup used to build while staging (each K / V element was converted N / 256 times per launch)
    // - and the ping-pong attention kernel stages rows by plain copy.  Needs the weight-fragment projection kernel (D = 256).  A sharing
    // layer refreshes only V and must keep the format of the cached q | k
    const bool img_ok = c->prec == 1 && c->kv_image && c->use_wf && D == 256 && L.proj_wf != nullptr;
    const bool kv_image = L.shared ? cache.kv_image : img_ok;
    if (L.shared && kv_image && !img_ok)
        return fail(IMP_E_STATE, "attention-sharing layer " + std::to_string(li) + ": the cached attention holds K / V as split-half images but the "
                                 "precision / kernel switches changed since; run the non-sharing layer of that kind again");
    const bool wf_proj = (wf && (c->use_wf > 1 || wf_tiles <= c->wf_proj_max_tiles)) || kv_image;
    const bool wf_mlp = wf && c->fuse_merge && cfg.norm_fn == IMP_NORM_IN && cfg.ac_fn == IMP_ACT_RELU && L.mlp0f_wf && L.mlp3_wf;
    // 1. projections: q|k|v of both images in one GEMM (the layer's weights are shared by the two images);
    //    a sharing layer only refreshes the value slot and keeps last iteration's q,k (== its probabilities)
    if (proj_done) {
        // (already in qkv[kind]: written by the chained launch of the previous layer)
    } else if (wf_proj && L.proj_wf) {
        WfParams p = wf_defaults();
        p.K = D; p.ksplit = D; p.N = L.proj.out; p.nside = 2;
        for (int s = 0; s < 2; ++s) {
            WfSide& g = p.side[s];
            g.A = desc[s]; g.M = n[s];
            g.C = L.shared ? qkv[s] + 2 * D : qkv[s];
            g.sA_b = (long)n[s] * D; g.sC_b = (long)n[s] * 3 * D;
        }
        p.Wf_ = L.proj_wf; p.bias = L.proj.b; p.lda = D; p.ldc = 3 * D;
        if (kv_image) p.kv_image_col = L.shared ? 0 : D;     // (a sharing layer's launch writes the value slot only)
        p.pass_split = wf_pass_split(c, wf_tiles, p.N);
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_wf(p, batch, st));
    } else {
        GemmParams p = gemm_defaults(c, D);
        for (int s = 0; s < 2; ++s) {
            GemmSide& g = p.side[s];
            g.A = desc[s]; g.W = L.proj.W; g.M = n[s]; g.N = L.proj.out;
            g.C = L.shared ? qkv[s] + 2 * D : qkv[s];
            g.sA_b = (long)n[s] * D; g.sC_b = (long)n[s] * 3 * D;
        }
        p.bias = L.proj.b; p.lda = D; p.ldw = D; p.ldc = 3 * D;
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_f32(p, batch, st));
    }
    if (c->rc.on && (kmask[0] || kmask[1])) return fail(IMP_E_ARG, "key masks and per-pair keypoint counts (imp_set_counts) exclude each other");
    if (!L.shared) {
        for (int img = 0; img < 2; ++img) {
            cache.masked[img] = kmask[img] != nullptr;
            if (kmask[img])
                HIP_TRY(hipMemcpyAsync(c->cmask[kind][img], kmask[img], (size_t)batch * n[img], hipMemcpyDeviceToDevice, st));
        }
        cache.valid = true; cache.batch = batch; cache.n[0] = n[0]; cache.n[1] = n[1];
        cache.kv_image = kv_image;
    }
    // 2. attention (nets/layers.py:121-131), sources: self -> same image, cross -> the other image
    {
        AttnParams a;
        memset(&a, 0, sizeof a);
        a.nside = 2; a.ldq = a.ldk = 3 * D; a.ldo = D; a.dh = c->dh;
        for (int s = 0; s < 2; ++s) {
            const int src = L.cross ? 1 - s : s;
            AttnSide& g = a.side[s];
            g.q = qkv[s]; g.k = qkv[src] + D; g.v = qkv[src] + 2 * D;
            g.out = c->attn_out[s];
            g.lse = L.shared ? nullptr : c->lse[kind][s];
            g.kmask = cache.masked[src] ? c->cmask[kind][src] : nullptr;
            g.sq_b = (long)n[s] * 3 * D; g.sk_b = (long)n[src] * 3 * D; g.so_b = (long)n[s] * D;
            g.nq = n[s]; g.nk = n[src];
            g.qimg = s; g.kimg = src;
        }
        a.rc = c->rc;
        a.kv_planes = kv_image ? 1 : 0;
        if (int arc = launch_attention(c, a, batch, st)) return arc;
    }
    // 3. merge conv (skipped when it is folded into mlp.0's weights)
    if (!c->fuse_merge) {
        const float* x[2] = {c->attn_out[0], c->attn_out[1]};
        int rc = linear_both(c, L.merge, batch, n, x, D, c->msg, D, st);
        if (rc) return rc;
    }
    const Linear& M0 = c->fuse_merge ? L.mlp0f : L.mlp0;
    // 4 + 5 FUSED (round 4): one launch for mlp.0 -> InstanceNorm -> ReLU -> mlp.3 (-> the next layer's projection) when every 64-row tile of
    // the launch gets a CU of its own (the tiles of an image wait for each other's statistics inside the kernel) and the launch is large
    // enough to pay for the wait; not under hipGraph capture (the exchange tags are launch parameters).  IMP_WF_FUSED=0 disables
    // ... and only while this stream has the device's waiting kernels to itself: sections of several streams (batch-steps in flight on replicas)
    // run strictly one after the other behind cross-stream events, and a waiting kernel that shares the chip with another stream's
    // ordinary kernel spins on the CUs it got until the rest of its workgroups find room - measured: three steps in flight with fused
    // layers 985 pairs/s, 1019 with the two-launch layers (which interleave freely), one step in flight 982 vs 957.  option wf_fused = 2: always
    if (wf_mlp && c->wf_fused && D == 256 && wf_tiles >= c->wf_fused_min_tiles && wf_tiles <= (long)c->num_cus && c->fx_rec[0] &&
        (c->wf_fused > 1 || !spin_gate_shared(c->device, st))) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            const bool can_chain = chain_li >= 0 && chain_li < (int)c->layers.size() && c->wf_chain && c->layers[chain_li].proj_wf && !kmask[0] && !kmask[1];
            const GnnLayer* NL = can_chain ? &c->layers[chain_li] : nullptr;
            const bool nimg = NL ? (NL->shared ? c->cache[NL->cross ? 1 : 0].kv_image : img_ok) : false;      // the format the next layer will expect
            if (int frc = launch_fused_layer(c, L, batch, n, desc, out, NL, NL ? c->qkv[NL->cross ? 1 : 0] : nullptr, nimg, st)) return frc;
            if (chained) *chained = can_chain;
            return IMP_OK;
        }
    }
    // 4. MLP conv 0 on cat([x, message]) (the concat is a K-split over two sources) + InstanceNorm statistics
    const bool in_norm = cfg.norm_fn == IMP_NORM_IN;
    const int maxn = n[0] > n[1] ? n[0] : n[1];
    int bm0 = gemm_stats_rows(maxn, 2 * D, 2 * batch);      // rows per statistics block of the MLP0 launch
    if (wf_mlp) {
        WfParams p = wf_defaults();
        p.K = 2 * D; p.ksplit = D; p.N = 2 * D; p.nside = 2;
        for (int s = 0; s < 2; ++s) {
            WfSide& g = p.side[s];
            g.A = desc[s]; g.A2 = c->attn_out[s]; g.C = c->hid[s]; g.M = n[s];
            g.sA_b = (long)n[s] * D; g.sA2_b = (long)n[s] * D; g.sC_b = (long)n[s] * 2 * D;
            g.out_stats = c->stats[s];
            g.fin_stats = c->nstat[s];                      // (mean, rstd) by the last workgroup to arrive: no finalize launch
        }
        p.stat_cnt = c->stat_cnt; p.norm_eps = 1e-3f;
        p.Wf_ = L.mlp0f_wf; p.bias = M0.b; p.lda = D; p.lda2 = D; p.ldc = 2 * D;
        p.pass_split = wf_pass_split(c, wf_tiles, p.N);
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_wf(p, batch, st));
    } else {
        GemmParams p = gemm_defaults(c, 2 * D);
        p.ksplit = D;
        for (int s = 0; s < 2; ++s) {
            GemmSide& g = p.side[s];
            g.A = desc[s]; g.A2 = c->fuse_merge ? c->attn_out[s] : c->msg[s]; g.W = M0.W; g.C = c->hid[s]; g.M = n[s]; g.N = 2 * D;
            g.sA_b = (long)n[s] * D; g.sC_b = (long)n[s] * 2 * D;
            g.out_stats = in_norm ? c->stats[s] : nullptr;
        }
        if (in_norm) p.flags |= GEMM_EPI_STATS;
        p.bias = M0.b; p.lda = D; p.lda2 = D; p.ldw = 2 * D; p.ldc = 2 * D;
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_f32(p, batch, st));
        if (in_norm) {
            StatsSide ss[2];
            for (int s = 0; s < 2; ++s) ss[s] = StatsSide{c->stats[s], c->nstat[s], (n[s] + bm0 - 1) / bm0, n[s], bm0};
            HIP_TRY(launch_stats_finalize(ss, 2, batch, 2 * D, 1e-3f, st, &c->rc));
        }
    }
    // 5. norm + activation while staging, conv 3, bias, residual add -> new descriptors; CHAIN: + the next layer's projection
    if (wf_mlp) {
        WfParams p = wf_defaults();
        p.K = 2 * D; p.ksplit = 2 * D; p.N = D; p.nside = 2;
        for (int s = 0; s < 2; ++s) {
            WfSide& g = p.side[s];
            g.A = c->hid[s]; g.C = out[s]; g.R = desc[s]; g.M = n[s];
            g.sA_b = (long)n[s] * 2 * D; g.sC_b = (long)n[s] * D; g.sR_b = (long)n[s] * D;
            g.in_stats = c->nstat[s];
        }
        p.norm_eps = 1e-3f;
        p.Wf_ = L.mlp3_wf; p.bias = L.mlp3.b; p.lda = 2 * D; p.ldc = D; p.ldr = D;
        p.pass_split = wf_pass_split(c, wf_tiles, p.N);
        const bool can_chain = chain_li >= 0 && chain_li < (int)c->layers.size() && c->wf_chain && D == 256 && wf_tiles >= c->wf_chain_min_tiles &&
                               c->layers[chain_li].proj_wf && !kmask[0] && !kmask[1];
        if (can_chain) {
            const GnnLayer& NL = c->layers[chain_li];
            float* const* nqkv = c->qkv[NL.cross ? 1 : 0];
            for (int s = 0; s < 2; ++s) {
                p.side[s].C2 = NL.shared ? nqkv[s] + 2 * D : nqkv[s];
                p.side[s].sC2_b = (long)n[s] * 3 * D;
            }
            p.Wf2_ = NL.proj_wf; p.bias2 = NL.proj.b; p.N2 = NL.proj.out; p.ldc2 = 3 * D;
            const bool nimg = NL.shared ? c->cache[NL.cross ? 1 : 0].kv_image : img_ok;      // the format the next layer will expect
            if (nimg) p.kv_image_col2 = NL.shared ? 0 : D;
            p.pass_split = 1;
            if (chained) *chained = true;
        }
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_wf(p, batch, st));
    } else {
        GemmParams p = gemm_defaults(c, 2 * D);
        p.flags = GEMM_PRO_NORM;
        p.act = cfg.ac_fn;
        if (!in_norm) {
            p.flags |= GEMM_PRO_AFFINE;
            p.nm_mean = L.bn.mean; p.nm_rstd = L.bn.rstd; p.nm_gamma = L.bn.gamma; p.nm_beta = L.bn.beta;
        }
        for (int s = 0; s < 2; ++s) {
            GemmSide& g = p.side[s];
            g.A = c->hid[s]; g.W = L.mlp3.W; g.C = out[s]; g.R = desc[s]; g.M = n[s]; g.N = D;
            g.sA_b = (long)n[s] * 2 * D; g.sC_b = (long)n[s] * D; g.sR_b = (long)n[s] * D;
            g.in_stats = in_norm ? c->nstat[s] : nullptr;
        }
        p.bias = L.mlp3.b; p.lda = 2 * D; p.ldw = 2 * D; p.ldc = D; p.ldr = D;
        apply_ragged(c, p);
        HIP_TRY(launch_gemm_f32(p, batch, st));
    }
    return IMP_OK;
}

int run_distance(imp_ctx* c, int layer_id, int batch, const int n[2], const float* const desc[2], float* dist,
                 hipStream_t st) {
    const int D = c->D;
    int idx = layer_id < 0 ? c->cfg.n_layers + layer_id : layer_id;
    if (idx < 0 || idx >= c->cfg.n_layers) return fail(IMP_E_ARG, "compute_distance: layer_id out of range");
    int rc = linear_both(c, c->final_proj[idx], batch, n, desc, D, c->mdesc, D, st);
    if (rc) return rc;
    GemmParams p = gemm_defaults(c, D);
    p.nside = 1;
    GemmSide& g = p.side[0];
    g.A = c->mdesc[0]; g.W = c->mdesc[1]; g.C = dist; g.M = n[0]; g.N = n[1];
    g.sA_b = (long)n[0] * D; g.sW_b = (long)n[1] * D; g.sC_b = (long)n[0] * n[1];
    p.lda = D; p.ldw = D; p.ldc = n[1];
    p.flags = GEMM_EPI_DIV;
    p.div = (float)std::sqrt((double)D);      // dist / descriptor_dim ** .5   nets/gm.py:294
    p.rc = c->rc; p.side[0].img = 0;          // (ragged batches: rows past a pair's own n0 are skipped; columns are computed to the padded n1 and never read)
    HIP_TRY(launch_gemm_f32(p, batch, st));
    return IMP_OK;
}

void ot_layout(imp_ctx* c, int n0, int n1, OtBuffers* o) {
    *o = c->ot;
    o->compact = c->ot_compact;
    o->ldp = (n1 + 1 + 3) & ~3;
    o->ldpt = (n0 + 1 + 3) & ~3;
}

// All resident launches of a device go through ONE stream: a resident kernel spins on group barriers, so two of them
// dispatched concurrently (pairs in flight on replicas, eval_loop workers) could each hold CUs the other is waiting for.
// Stream order on the lane rules that out; the caller's stream is joined in and out with events.
struct ResidentLane { std::mutex mu; hipStream_t stream = nullptr; };
ResidentLane* resident_lane(int device) {
    static std::mutex mu;
    static std::map<int, ResidentLane*> lanes;
    std::lock_guard<std::mutex> lock(mu);
    auto it = lanes.find(device);
    if (it != lanes.end()) return it->second;
    ResidentLane* l = new ResidentLane();
    if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) { delete l; return nullptr; }
    lanes[device] = l;
    return l;
}

// SPIN GATE (round 4).  Two kinds of kernels of this library WAIT inside the launch for other workgroups of the same launch: the chip-
// resident Sinkhorn and the fused layer MLP (gemm_wf.hip).  Each needs all its workgroups co-resident, so two of them must never be
// dispatched side by side (each could hold CUs the other is waiting for): every such launch runs inside a gate section of its device.
// The gate orders the sections in the order the host enters them: a section on stream B first waits (hipStreamWaitEvent) for the event
// the previous section left on stream A.  While all sections come from ONE stream (the common case: one context, one stream) stream
// order already serialises them and the gate records nothing - no event, no extra packet in the stream; the first section from another
// stream switches the gate to recording mode (one event per section) after ordering itself behind the first stream's tail.
struct SpinGate {
    std::mutex mu;
    hipEvent_t ring[64] = {};
    int head = 0, same_run = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t last_event = nullptr;
    bool has_last = false, multi = false;
    // the kernel-choice hint (spin_gate_shared) keeps its own history of who ASKED: the two-launch layers it selects enter no section, so
    // with sections alone a stream that took over the device (a new pipeline, the bench's one-in-flight leg) saw "shared" until seven of its
    // Sinkhorn launches - seven whole steps - had passed (round 5: found in a kernel trace, tools/trace_sequence.py)
    hipStream_t last_query = nullptr;
    bool has_query = false, query_multi = false;
    int query_run = 0;                                       // consecutive queries of one stream
    std::chrono::steady_clock::time_point query_switch;      // (option fused_choice = 1) when a query last came from another stream than the one before
};
SpinGate* spin_gate(int device) {
    static std::mutex mu;
    static std::map<int, SpinGate*> gates;
    std::lock_guard<std::mutex> lock(mu);
    auto it = gates.find(device);
    if (it != gates.end()) return it->second;
    SpinGate* g = new SpinGate();
    for (auto& e : g->ring)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete g; return nullptr; }
    gates[device] = g;
    return g;
}
// do sections of other streams alternate with `st`'s on this device at the moment?  (a hint for choosing kernels, not a guarantee)
static int imp_fused_choice_by_time = 0;
static int imp_fused_alone_after = 6;      // a stream counts as alone once MORE than this many kernel choices in a row were its own (option fused_alone_after)          // A/B hook (option fused_choice = 1): the rule the round-5 advisor proposed, below
bool spin_gate_shared(int device, hipStream_t st) {
    SpinGate* g = spin_gate(device);
    if (!g) return false;
    std::lock_guard<std::mutex> lock(g->mu);
    // "shared" = another stream is choosing kernels right now.  A stream counts as alone again after seven queries in a row (round 5).  One imp_match_pair
    // asks 18 times inside a single call, so with three steps in flight a stream whose neighbours are blocked in a synchronisation flips to "alone" in
    // mid-call and takes the fused launch for the rest of it - ADVICE r5 called that a risk and proposed a rule by time (alone = nobody else asked for
    // 8 ms).  Round 6 built it and measured it: with three steps in flight it never takes the fused launch (1 040-1 062 pairs/s on three boxes against
    // 1 085-1 098 with this rule; round 5's own A/B: mixed 1 121, always fused 1 094, never fused 1 065 - profiles/r05/envab_fused_policy_three_in_flight.log).
    // The mix IS the optimum: the fused launch where a stream happens to have the chip, the two-launch layers where it shares it; the gate's sections keep
    // it correct either way.  The count stays; the rule by time is kept behind option fused_choice = 1 for the A/B (profiles/r06/fused_choice_ab.log)
    if (imp_fused_choice_by_time) {
        const auto now = std::chrono::steady_clock::now();
        if (g->has_query && g->last_query != st) { g->query_multi = true; g->query_switch = now; }
        else if (g->query_multi && now - g->query_switch > std::chrono::milliseconds(8)) g->query_multi = false;
    } else {
        if (g->has_query && g->last_query != st) { g->query_multi = true; g->query_run = 0; }
        else if (g->query_multi && ++g->query_run > imp_fused_alone_after) g->query_multi = false;
    }
    g->last_query = st; g->has_query = true;
    return g->query_multi;
}
// enters a gate section on `st` (the mutex stays locked until spin_leave): everything enqueued on `st` from here on runs after the
// previous section of the device has finished
int spin_enter(SpinGate* g, hipStream_t st) {
    g->mu.lock();
    if (g->has_last && g->last_stream != st) {
        hipError_t e = hipSuccess;
        if (!g->multi) {
            // the earlier sections left no event: one at the tail of their stream is late (conservative), never early
            g->multi = true;
            hipEvent_t ev = g->ring[g->head++ & 63];
            e = hipEventRecord(ev, g->last_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
            else e = hipDeviceSynchronize();             // (that stream no longer exists: wait for whatever is left of it)
        } else {
            e = hipStreamWaitEvent(st, g->last_event, 0);
        }
        g->same_run = 0;
        if (e != hipSuccess) { g->mu.unlock(); return fail(IMP_E_HIP, std::string("spin gate: ") + hipGetErrorString(e)); }
    } else if (g->multi && ++g->same_run > 6) {
        g->multi = false;                                // one stream again for a long time: stop recording
    }
    return IMP_OK;
}
int spin_leave(SpinGate* g, hipStream_t st) {
    hipError_t e = hipSuccess;
    if (g->multi) {
        hipEvent_t ev = g->ring[g->head++ & 63];
        e = hipEventRecord(ev, st);
        g->last_event = ev;
    }
    g->last_stream = st;
    g->has_last = true;
    g->mu.unlock();
    if (e != hipSuccess) return fail(IMP_E_HIP, std::string("spin gate: ") + hipGetErrorString(e));
    return IMP_OK;
}

constexpr size_t kResidentMaxLdx = 256 * 16 + 4;
int ensure_resident_buffers(imp_ctx* c, int batch) {
    if (c->xpart && batch <= c->xcap_b) return IMP_OK;
    const size_t wgs = (size_t)c->num_cus;
    int rc = 0;
    const int cap = batch < 8 ? 8 : batch;
    if (!c->xpart) {
        rc = dev_alloc(c, c->allocs_x, &c->xpart, 2 * wgs * kResidentMaxLdx);
        if (!rc) HIP_TRY(hipMemset(c->xpart, 0, 2 * wgs * kResidentMaxLdx * sizeof(float)));     // tag 0 = never written
        if (!rc) rc = dev_alloc(c, c->allocs_x, &c->xmax, 4 * wgs * kResidentMaxLdx);
        if (!rc) HIP_TRY(hipMemset(c->xmax, 0, 4 * wgs * kResidentMaxLdx * sizeof(float)));
        if (!rc) rc = dev_alloc(c, c->allocs_x, &c->xstatus, 32);
        if (!rc) HIP_TRY(hipMemset(c->xstatus, 0, 128));
        if (!rc) { const unsigned graph_tags = c->graph_tag0; HIP_TRY(hipMemcpy(c->xstatus + 16, &graph_tags, 4, hipMemcpyHostToDevice)); }   // graph launches tag in the upper half
        c->ticket_base = 0;
        if (!rc) HIP_TRY(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        if (!rc) HIP_TRY(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    if (!rc) rc = dev_alloc(c, c->allocs_x, &c->xv, 2 * (size_t)cap * kResidentMaxLdx);
    if (!rc) HIP_TRY(hipMemset(c->xv, 0, 2 * (size_t)cap * kResidentMaxLdx * sizeof(float)));
    if (!rc) rc = dev_alloc(c, c->allocs_x, &c->xhalf, 4 * (size_t)cap * kResidentMaxLdx);
    if (!rc) HIP_TRY(hipMemset(c->xhalf, 0, 4 * (size_t)cap * kResidentMaxLdx * sizeof(float)));
    if (rc) return rc;
    // the clears above run on the NULL stream, which the callers' (non-blocking) streams and the lane do not wait for:
    // they must have landed before the first resident kernel writes its tags into these buffers
    HIP_TRY(hipDeviceSynchronize());
    c->xcap_b = cap;
    return IMP_OK;
}

// reserves the tags of one launch.  Tags must never repeat on the exchange buffers; when the 32-bit counter is about to
// wrap (after ~20 million launches) the buffers are cleared and the count restarts.
unsigned resident_tags(imp_ctx* c, int iterations) {
    const unsigned need = 3u * (unsigned)iterations + 4u;
    if (c->xtag > 0x7FFFFFFFu - need - 8u) {           // (the upper half of the tag space belongs to launches replayed from hipGraphs)
        (void)hipDeviceSynchronize();
        const size_t wgs = (size_t)c->num_cus;
        (void)hipMemset(c->xpart, 0, 2 * wgs * kResidentMaxLdx * sizeof(float));
        (void)hipMemset(c->xmax, 0, 4 * wgs * kResidentMaxLdx * sizeof(float));
        (void)hipMemset(c->xv, 0, 2 * (size_t)c->xcap_b * kResidentMaxLdx * sizeof(float));
        (void)hipMemset(c->xhalf, 0, 4 * (size_t)c->xcap_b * kResidentMaxLdx * sizeof(float));
        c->xtag = 0;
    }
    const unsigned base = c->xtag;
    c->xtag += need;
    return base;
}


// health plumbing of a resident launch: the mapped host word, and - LOCAL launches - the per-XCC ticket counters with the value
// they hold before this launch (every XCC receives G * ceil(B / 8) resp. G / 2 workgroups of it)
void resident_health_params(imp_ctx* c, OtResidentParams* p) {
    p->host_status = c->xstatus_hostdev;
    p->xcc_tickets = reinterpret_cast<unsigned*>(c->xstatus) + 4;
    p->ticket_base = c->ticket_base;
    p->fake_placement = c->ot_fake;
    if (p->local == 1) c->ticket_base += (unsigned)(p->G * ((p->B + 7) / 8));
    else if (p->local == 2) c->ticket_base += (unsigned)(p->G / 2);
}

// Non-synchronising health check, run at every compute entry point (check_ready): has a resident launch of this context
// timed out since the last look?  Then the results of that EARLIER call are void (the kernel poisoned them: NaN scores, no
// matches).  The context recovers - waits for its work, resets the exchange state, stops using the protocol that failed
// (first the XCD-local launches, then the resident kernel altogether) - and the entry point that noticed returns
// IMP_E_RESIDENT so that the caller re-runs the voided batch.
int resident_health(imp_ctx* c) {
    if (!c->xstatus_host) return IMP_OK;
    // the resident word first: a voided launch leaves NaN maxima behind, which the match kernel also reports through the range word -
    // that is one event (IMP_E_RESIDENT), not two
    const int st = *static_cast<volatile int*>(c->xstatus_host);
    if (st == 0 && static_cast<volatile int*>(c->xstatus_host)[2]) {
        // not a failure: the device-side tag counter of hipGraph-replayed resident launches wrapped (ot_resident.hip leave()).  Their tags
        // start over, so the exchange buffers must not hold old ones: wait for the device and clear them
        (void)hipSetDevice(c->device);
        if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return IMP_OK; }      // (e.g. a capture in progress: at the next entry point)
        const size_t wgs = (size_t)c->num_cus;
        if (c->xpart) (void)hipMemset(c->xpart, 0, 2 * wgs * kResidentMaxLdx * sizeof(float));
        if (c->xmax) (void)hipMemset(c->xmax, 0, 4 * wgs * kResidentMaxLdx * sizeof(float));
        if (c->xv) (void)hipMemset(c->xv, 0, 2 * (size_t)c->xcap_b * kResidentMaxLdx * sizeof(float));
        if (c->xhalf) (void)hipMemset(c->xhalf, 0, 4 * (size_t)c->xcap_b * kResidentMaxLdx * sizeof(float));
        c->xtag = 0;                                       // (the eager launches' tags may start over as well: the buffers are clean)
        (void)hipDeviceSynchronize();
        static_cast<volatile int*>(c->xstatus_host)[2] = 0;
        c->tag_wraps += 1;
        return IMP_OK;
    }
    if (st) {
        (void)hipSetDevice(c->device);                     // (the caller's thread may have another device current: imp_resident_health)
        (void)hipDeviceSynchronize();
        {   // post-mortem (round 6): the first timed-out waiter's record + the host's view at the moment it noticed
            volatile int* pm = static_cast<volatile int*>(c->xstatus_host) + IMP_PM_BASE;
            for (int i = 0; i < 15; ++i) c->postmortem[i] = pm[i];
            for (int i = 0; i < 15; ++i) pm[i] = 0;
            SpinGate* g = spin_gate(c->device);
            c->postmortem[16] = st;
            c->postmortem[17] = g ? (g->multi ? 1 : 0) : -1;              // gate was recording cross-stream events (several streams took sections)
            c->postmortem[18] = g ? (g->query_multi ? 1 : 0) : -1;        // several streams were choosing kernels
            c->postmortem[19] = g ? g->same_run : -1;
            c->postmortem[20] = c->ot_degrade;
            c->postmortem[21] = c->wf_fused;
            c->postmortem[22] = (int)c->xtag;                             // next Sinkhorn tag base / last fused tag: places the voided launch in the context's sequence
            c->postmortem[23] = (int)c->fx_tag;
            c->postmortem[24] = c->resident_timeouts + 1;
            c->postmortem_valid = 1;
        }
        if (c->xstatus) (void)hipMemset(c->xstatus, 0, 64);
        if (c->xstatus) (void)hipMemset(c->xstatus + 17, 0, 44);           // graph launches: ticket base, done counter, tickets (their tag base [16] keeps counting)
        if (c->fx_status) (void)hipMemset(c->fx_status, 0, 64);
        (void)hipDeviceSynchronize();
        *static_cast<volatile int*>(c->xstatus_host) = 0;
        *static_cast<volatile int*>(c->range_host) = 0;    // (raised by the match kernel that met the poisoned maxima)
        c->ticket_base = 0;
        c->resident_timeouts += 1;
        if (st == 3) {
            // a fused layer launch (gemm_wf.hip gemm_wf_fused_kernel) timed out in its statistics exchange: the descriptors it wrote are NaN
            c->wf_fused = 0;
            return fail(IMP_E_RESIDENT, "a fused GNN-layer launch of this context timed out in its InstanceNorm statistics exchange: the results of "
                                        "that call are void (scores NaN, no matches); the context now runs the layer's MLP as two launches - re-run the batch. "
                                        "Every call on this context since its last health check is void");
        }
        const int was = c->ot_degrade;
        c->ot_degrade = was < 2 ? was + 1 : 2;
        return fail(IMP_E_RESIDENT, std::string("a chip-resident Sinkhorn launch of this context ") +
                                        (st == 2 ? "was not spread evenly over the XCCs" : "timed out in an exchange") +
                                        ": the results of that call - and of every call on this context since its last health check - are void "
                                        "(scores NaN, no matches); the context now uses " +
                                        (c->ot_degrade == 1 ? "the chip-wide exchange only" : "the streaming kernels") + " - re-run the batch");
    }
    if (*static_cast<volatile int*>(c->range_host)) {
        // a match kernel of an EARLIER call met non-finite score maxima: in f16x3 mode every MFMA operand must stay inside the fp16
        // range (|x| < 65504: hi = f16(x) overflows to inf beyond it and the product turns into NaN); non-finite INPUTS look the same
        *static_cast<volatile int*>(c->range_host) = 0;
        c->range_events += 1;
        return fail(IMP_E_RANGE, c->prec == 1
                    ? "non-finite match scores in an earlier call on this context: an operand left the fp16 range of the split-half f16x3 "
                      "arithmetic (|x| >= 65504) or the inputs were not finite - that call's matches are void (all -1); use precision = 'f32' "
                      "(imp_set_precision(ctx, 0)) for such data"
                    : "non-finite match scores in an earlier call on this context (non-finite inputs or weights): that call's matches are void");
    }
    return IMP_OK;
}

// the resident launch on the device's lane, joined to `st` on both sides; returns IMP_OK, or >0 when not applicable
// chooses the decomposition of a resident launch.  XCD-local (every pair on the 32 CUs of one XCD, exchanges through that XCD's
// L2 instead of across the fabric) whenever a pair fits there: option ot_local = 0 disables.  Returns 0 when nothing fits.
int plan_resident(imp_ctx* c, int batch, int n0, int n1, int max_wgs, int* nch, int* rpw, int* G, int* local) {
    // XCD-local protocols: only on the layout they were written for (8 XCCs x 32 CUs, SPX) and while no launch of this context
    // ever timed out (ot_degrade)
    const bool hw_local = c->num_xccs == 8 && c->num_cus == 256 && c->ot_degrade == 0;
    const bool allow_local = c->ot_local != 0 && hw_local;
    // (round 6) WHICH decomposition a launch gets may depend on the batch - the column sums come out bit-identical in all of them (the canonical tree
    // of ot_resident.hip) - with one exception: the two wide shapes (n1 > 2048) sum in another order and are planned only for ONE pair at its own sizes
    const int single = batch == 1 && !(c->rc.on && (c->rc.n[0][0] != n0 || c->rc.n[1][0] != n1));
    *local = 0;
    const int per_xcd = (batch + 7) / 8;
    if (allow_local && max_wgs >= c->num_cus && per_xcd <= 32 && ot_resident_plan(1, n0, n1, 32 / per_xcd, nch, rpw, G, single)) {
        *local = 1;
        return 1;
    }
    // two XCDs per pair: one fabric crossing per iteration instead of two (ot_resident.hip, LOCAL = 2).  Always 64 workgroups per pair - the halves
    // meet at row 1024, a node of the tree, whatever n0 is (workgroups past n0 hold no rows and still own a column slice of the exchange)
    if (c->ot_hier && hw_local && max_wgs >= c->num_cus && batch <= 4 && ot_resident_plan(1, n0, n1, 64, nch, rpw, G, 0)) {
        *G = 64;
        if (ot_resident_hier_ok(*nch, *rpw, *G, batch)) { *local = 2; return 1; }
    }
    return ot_resident_plan(batch, n0, n1, max_wgs, nch, rpw, G, single);
}

// one resident launch over the pairs [b0, b0 + nb) of a batch (all per-pair arrays are indexed b * stride inside the kernel)
int run_score_resident_launch(imp_ctx* c, int batch, int b0, int nb, int n0, int n1, const float* dist, float bin, int iterations, float* scores,
                              bool want_max, bool want_uv, int nch, int rpw, int G, int local, hipStream_t st) {
    ResidentLane* lane = c->ot_lane ? resident_lane(c->device) : nullptr;
    if (c->ot_lane && !lane) return 1;
    OtResidentParams p;
    memset(&p, 0, sizeof p);
    p.dist = dist + (size_t)b0 * n0 * n1; p.B = nb; p.n0 = n0; p.n1 = n1; p.T = iterations; p.G = G; p.bin = bin;
    p.xpart = c->xpart; p.xv = c->xv; p.xmax = c->xmax; p.status = c->xstatus;
    p.local = local; p.xhalf = c->xhalf;
    p.rc = c->rc;
    p.tag_base = resident_tags(c, iterations);
    resident_health_params(c, &p);
    if (want_uv) {
        p.ldu = (n0 + 1 + 3) & ~3; p.ldv = (n1 + 1 + 3) & ~3;
        p.u = c->ot.u + (size_t)b0 * p.ldu; p.v = c->ot.v + (size_t)b0 * p.ldv;
    }
    p.scores = scores ? scores + (size_t)b0 * (n0 + 1) * (n1 + 1) : nullptr;
    if (want_max) {
        p.max0 = c->max0 + (size_t)b0 * n0; p.arg0 = c->arg0 + (size_t)b0 * n0;
        p.max1 = c->max1 + (size_t)b0 * n1; p.arg1 = c->arg1 + (size_t)b0 * n1;
    }
    SpinGate* gate = spin_gate(c->device);
    if (!gate) return 1;
    if (int grc = spin_enter(gate, st)) return grc;
    hipError_t e = hipSuccess;
    if (c->ot_lane) {
        // rounds 2-3: the launch on the device's lane stream, joined to `st` on both sides (two cross-stream event hops per launch)
        std::lock_guard<std::mutex> lock(lane->mu);
        e = hipEventRecord(c->ev_in, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(lane->stream, c->ev_in, 0);
        if (e == hipSuccess) e = launch_ot_resident(p, nch, rpw, lane->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev_out, lane->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, c->ev_out, 0);
    } else {
        e = launch_ot_resident(p, nch, rpw, st);
        if (e == hipSuccess && c->ot_verify) e = hipEventRecord(c->ev_out, st);      // (verify mode waits on ev_out)
    }
    const int lrc = spin_leave(gate, st);
    HIP_TRY(e);
    return lrc;
}

// A resident launch recorded into a hipGraph: on the capturing stream itself (no lane: the replay knows no host mutex), with the tag and
// ticket bases in device memory (OtResidentParams::dev_base) so that every replay exchanges under fresh tags.  Nothing may be allocated
// or synchronised under capture: the buffers must exist (one ordinary call before the capture), else the streaming kernels are recorded.
// A graph that holds such a launch must not be replayed while another resident launch of the process runs (another context, another
// graph): nothing serialises them, a collision shows as a time-out (NaN outputs, IMP_E_RESIDENT at the next entry point) - detected, not prevented.
int run_score_resident_graph(imp_ctx* c, int batch, int n0, int n1, const float* dist, float bin, int iterations, float* scores,
                             bool want_max, bool want_uv, hipStream_t st) {
    if (!c->xpart || batch > c->xcap_b || c->ot_degrade >= 2 || c->ot_graph == 0) return 1;
    int nch, rpw, G, local;
    if (!plan_resident(c, batch, n0, n1, c->num_cus, &nch, &rpw, &G, &local)) return 1;
    OtResidentParams p;
    memset(&p, 0, sizeof p);
    p.dist = dist; p.B = batch; p.n0 = n0; p.n1 = n1; p.T = iterations; p.G = G; p.bin = bin;
    p.xpart = c->xpart; p.xv = c->xv; p.xmax = c->xmax; p.status = c->xstatus;
    p.local = local; p.xhalf = c->xhalf;
    p.rc = c->rc;
    p.host_status = c->xstatus_hostdev;
    p.dev_base = reinterpret_cast<unsigned*>(c->xstatus) + 16;
    p.xcc_tickets = reinterpret_cast<unsigned*>(c->xstatus) + 20;
    p.fake_placement = c->ot_fake;
    if (want_uv) { p.ldu = (n0 + 1 + 3) & ~3; p.ldv = (n1 + 1 + 3) & ~3; p.u = c->ot.u; p.v = c->ot.v; }
    p.scores = scores;
    if (want_max) { p.max0 = c->max0; p.arg0 = c->arg0; p.max1 = c->max1; p.arg1 = c->arg1; }
    HIP_TRY(launch_ot_resident(p, nch, rpw, st));
    return IMP_OK;
}

int run_score_resident(imp_ctx* c, int batch, int n0, int n1, const float* dist, float bin, int iterations, float* scores,
                       bool want_max, bool want_uv, hipStream_t st) {
    if (!c->ot_resident) return 1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) return 1;
    if (cap != hipStreamCaptureStatusNone) return run_score_resident_graph(c, batch, n0, n1, dist, bin, iterations, scores, want_max, want_uv, st);
    int rc = ensure_resident_buffers(c, batch);
    if (rc) return rc;
    for (;;) {
        if (c->ot_degrade >= 2) return 1;                                  // a chip-wide launch timed out before: streaming kernels only
        int nch, rpw, G, local;
        if (!plan_resident(c, batch, n0, n1, c->num_cus, &nch, &rpw, &G, &local)) return 1;
        rc = run_score_resident_launch(c, batch, 0, batch, n0, n1, dist, bin, iterations, scores, want_max, want_uv, nch, rpw, G, local, st);
        if (rc || !c->ot_verify) return rc;
        // verified mode: wait for the launch; a voided one is re-run inside this call on the next protocol down
        HIP_TRY(hipEventSynchronize(c->ev_out));
        const int hrc = resident_health(c);
        if (hrc != IMP_E_RESIDENT) return hrc;                 // IMP_OK, or an event of another kind (IMP_E_RANGE of an earlier call): the caller's
    }
}

// scores (optional) and, with max_done, the row / column maxima of the inner block in c->max0/arg0/max1/arg1.
// *max_done is set when the maxima were produced here (resident path); otherwise the caller takes them from ot_out.
int run_score(imp_ctx* c, int batch, int n0, int n1, const float* dist, float bin, int iterations, int with_sinkhorn,
              float* scores, OtBuffers* ot_out, hipStream_t st, bool* max_done = nullptr) {
    OtBuffers o;
    ot_layout(c, n0, n1, &o);
    const int dual = with_sinkhorn ? 0 : 1;
    if (max_done) *max_done = false;
    // (a one-pair "ragged" batch whose counts equal the padded sizes IS a uniform batch - the streaming kernels below take it, so a
    // single pair always runs, also on a context that has stepped down from the resident kernel or past its size limits)
    const bool ragged = c->rc.on && !(batch == 1 && c->rc.n[0][0] == n0 && c->rc.n[1][0] == n1);
    if (ragged && (dual || !max_done))
        return fail(IMP_E_ARG, "ragged batches (imp_set_counts) take the fused score + matches path only: Sinkhorn scorer (imp_match_pair / imp_match_tail)");
    if (!dual && (scores || max_done)) {
        const int rr = run_score_resident(c, batch, n0, n1, dist, bin, iterations, scores, max_done != nullptr, true, st);
        if (rr < 0) return rr;
        if (rr == 0) {
            if (max_done) *max_done = true;
            if (ot_out) *ot_out = o;
            return IMP_OK;
        }
    }
    if (ragged)
        return fail(IMP_E_NOFIT, "ragged batch beyond the chip-resident Sinkhorn kernel (sizes, or the context fell back to the streaming kernels): run these pairs one call each");
    HIP_TRY(launch_ot_init(dist, batch, n0, n1, bin, dual, o, st));
    if (dual) HIP_TRY(launch_ot_dual_lse(batch, n0, n1, o, st));
    else HIP_TRY(launch_ot_iterations(batch, n0, n1, iterations, o, st));
    if (scores) HIP_TRY(launch_ot_scores(batch, n0, n1, dual, o, scores, st));
    if (ot_out) *ot_out = o;
    return IMP_OK;
}

// K of the cached attention `kind`, image `side`, as fp32 rows: the k slot itself (*ld = 3 D), or - when the projection wrote split-half
// images - their exact fp32 value hi + lo converted into the context's scratch (*ld = D).  (The attention kernel multiplied exactly these
// values: readers that recompute probabilities from q, k and the stored log-sum-exp see the operands it saw.)
const float* cached_k_fp32(imp_ctx* c, int kind, int side, long* ld, hipStream_t st, hipError_t* err) {
    const AttnCache& cache = c->cache[kind];
    const int D = c->D;
    *err = hipSuccess;
    if (!cache.kv_image) { *ld = 3 * D; return c->qkv[kind][side] + D; }
    *err = launch_attn_kv_unplanes(c->qkv[kind][side], (long)cache.batch * cache.n[side], 3 * D, D, c->dh, c->kf32[side], D, st);
    *ld = D;
    return c->kf32[side];
}

// TEST HOOK (imp_debug_hold_cus): `workgroups` workgroups that each keep 96 KB of LDS - no waiting kernel of this library fits beside one on a CU - for
// `ticks` ticks of the constant 100 MHz counter: what an RCCL kernel waiting for a slower peer looks like to the launches that need every CU
__global__ __launch_bounds__(256) void hold_cus_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold_lds[];
    hold_lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && hold_lds[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) *sink = 1;        // (keeps the LDS allocation alive)
}

struct ProbSpec { int kind, qside, kside; };
// which: 0 self img0, 1 self img1, 2 cross img0<-img1 (reference cross_prob1), 3 cross img1<-img0 (cross_prob0)
const ProbSpec kProb[4] = {{0, 0, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 0}};

__global__ void zero_masked_columns_kernel(float* prob, const uint8_t* mask, int nq, int nk, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int key = (int)(i % nk);
    const long b = i / ((long)IMP_NUM_HEADS * nq * nk);
    if (!mask[b * nk + key]) prob[i] = 0.f;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// Host threads for the pose estimates of the native lock-step loop: imp_estimate_pose keeps a per-THREAD workspace (device buffers, a
// pinned staging page, the MAGSAC++ weight table), so the workers live as long as the context; each owns a stream.
struct PoseWorkers {
    std::vector<std::thread> threads;
    std::deque<std::function<void(hipStream_t)>> jobs;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    int device = 0;
    explicit PoseWorkers(int n, int dev) : device(dev) {
        for (int i = 0; i < n; ++i)
            threads.emplace_back([this] {
                (void)hipSetDevice(device);
                hipStream_t st = nullptr;
                (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                for (;;) {
                    std::function<void(hipStream_t)> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return stop || !jobs.empty(); });
                        if (stop && jobs.empty()) break;
                        job = std::move(jobs.front());
                        jobs.pop_front();
                    }
                    job(st);
                }
                if (st) (void)hipStreamDestroy(st);
            });
    }
    void submit(std::function<void(hipStream_t)> f) {
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(f)); }
        cv.notify_one();
    }
    ~PoseWorkers() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : threads) t.join();
    }
};

namespace {
void loop_release(imp_ctx* c) {
    delete c->lp_workers; c->lp_workers = nullptr;
    if (c->lp_idx) (void)hipFree(c->lp_idx);
    if (c->lp_ms) (void)hipFree(c->lp_ms);
    if (c->lp_pin) (void)hipHostFree(c->lp_pin);
    if (c->lp_ev) (void)hipEventDestroy(c->lp_ev);
    c->lp_idx = nullptr; c->lp_ms = nullptr; c->lp_pin = nullptr; c->lp_ev = nullptr; c->lp_cap = 0;
    if (c->lu_scores) (void)hipFree(c->lu_scores);
    if (c->lu_pool) (void)hipFree(c->lu_pool);
    if (c->lu_pool_pin) (void)hipHostFree(c->lu_pool_pin);
    for (int s = 0; s < 2; ++s) { if (c->lu_alt[s]) (void)hipFree(c->lu_alt[s]); c->lu_alt[s] = nullptr; c->lu_alt_cap[s] = 0; }
    c->lu_scores = nullptr; c->lu_pool = nullptr; c->lu_pool_pin = nullptr; c->lu_scores_cap = 0; c->lu_pool_cap = 0;
}
// rotation angle between two rotation matrices / angle between two vectors, degrees (imp_release_amd/matching.py angle_error_mat / _vec,
// the semantics of tools/utils.py:425-431)
double loop_angle_mat(const double* R1, const double* R2) {
    double tr = 0.0;                                    // trace(R1^T R2) = sum_ij R1[i][j] R2[i][j]
    for (int i = 0; i < 9; ++i) tr += R1[i] * R2[i];
    double cs = (tr - 1.0) / 2.0;
    cs = cs < -1.0 ? -1.0 : (cs > 1.0 ? 1.0 : cs);
    return std::fabs(std::acos(cs)) * 180.0 / M_PI;
}
double loop_angle_vec(const double* a, const double* b) {
    const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    double cs = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) / n;
    cs = cs < -1.0 ? -1.0 : (cs > 1.0 ? 1.0 : cs);
    return std::acos(cs) * 180.0 / M_PI;
}
struct LoopPose {                                       // one pose estimate in flight
    std::vector<float> k0, k1;
    std::vector<unsigned char> mask;
    double E[9], R[9], t[3];
    int n = 0, rc = 1, ninl = 0;
    std::promise<void> done;
    std::future<void> fut;
};
struct LoopPair {
    bool live = true, has_last = false, scored = false;
    double lastR[9], lastT[3];
    std::shared_ptr<LoopPose> pend;
    int pend_it = -1;
    std::vector<int> pm0, pm1;                          // matches handed to the pose step of the pending iteration
    std::vector<int64_t> pend_idx, last_idx;
    std::vector<float> pend_ms, last_ms;
};
}  // namespace

// ================================================================================================
int imp_fail(int code, const char* msg) { return fail(code, msg); }   // for the other translation units (superpoint.hip)

extern "C" {

const char* imp_last_error(void) { return g_err.c_str(); }
const char* imp_version(void) { return "imp_hip 0.3 gfx950 f16x3-mfma|f32-mfma"; }

// named switches of a context (include/imp_hip.h imp_ctx_option): the step-down paths a context takes by itself after a voided waiting launch, forced for A/B tests,
// and the fault-injection hooks of the test suite.  Set before the first compute call.
static int ctx_option(imp_ctx* c, const char* name, long v) {
    const std::string n(name ? name : "");
    if (n == "ot_resident") c->ot_resident = v != 0;                  // 0: streaming Sinkhorn kernels (ot.hip)
    else if (n == "ot_local") c->ot_local = v != 0;                   // 0: no XCD-local decomposition of the resident kernel
    else if (n == "ot_hier") c->ot_hier = v != 0;                     // 0: no two-XCDs-per-pair decomposition
    else if (n == "ot_verify") c->ot_verify = v != 0;                 // 1: every resident launch is waited for and repaired inside the call
    else if (n == "ot_graph") c->ot_graph = v != 0;                   // 0: hipGraph captures record the streaming Sinkhorn
    else if (n == "ot_fake_placement") c->ot_fake = v != 0;           // TEST HOOK: XCD-local workgroups report a wrong XCC
    else if (n == "ot_graph_tag0") { if ((unsigned long)v >= 0x80000000ul && (unsigned long)v <= 0xFFFFF000ul) c->graph_tag0 = (unsigned)v; }   // TEST HOOK: tag wrap-around
    else if (n == "gemm_wf") c->use_wf = (int)v;                      // 0: plain tiled GEMMs, 1: weight-fragment GEMMs for large launches, 2: always
    else if (n == "wf_chain") c->wf_chain = (int)v;                   // 0: MLP3 and the next projection as two launches
    else if (n == "wf_chain_min") c->wf_chain_min_tiles = v;
    else if (n == "wf_fused") c->wf_fused = (int)v;                   // 0: the layer's MLP as two launches (no in-kernel statistics exchange)
    else if (n == "wf_fused_min") c->wf_fused_min_tiles = v;
    else if (n == "wf_fused_fake") c->wf_fused_fake = v != 0;         // TEST HOOK: one workgroup withholds its statistics
    else if (n == "probe_prof") c->probe_prof = v != 0;               // probes: the timing entry points also print the phase cycle stamps of a profiling build
    else if (n == "fused_choice") imp_fused_choice_by_time = v != 0;  // A/B: 1 = a stream is alone when nobody else asked for 8 ms (process-wide; measured slower: spin_gate_shared)
    else if (n == "fused_alone_after") imp_fused_alone_after = (int)v; // sweep hook for the count rule (process-wide)
    else if (n == "attn_shares") c->attn_share_mode = (int)v;          // key shares of a split attention unit: 0 launcher's choice, 1 one workgroup each, 2 one workgroup all (same bits)
    else if (n == "kv_image") c->kv_image = (int)v;                   // 0: the projection writes fp32 k | v, the attention kernel splits them while staging
    else return IMP_E_ARG;
    return IMP_OK;
}

int imp_create(imp_ctx** out, const imp_config* cfg, int device) {
    if (!out || !cfg) return fail(IMP_E_ARG, "imp_create: null argument");
    if (cfg->descriptor_dim != 256 && cfg->descriptor_dim != 128)
        return fail(IMP_E_ARG, "descriptor_dim must be 256 or 128 (4 heads x 64 / 32)");
    if (cfg->n_gnn_layers < 0 || cfg->n_gnn_layers > 128 || cfg->n_layers < 1)
        return fail(IMP_E_ARG, "bad layer counts");
    const int nk = n_kenc(*cfg);
    if (nk < 1 || cfg->kenc_channels[0] > 64) return fail(IMP_E_ARG, "keypoint_encoder needs >= 1 layer, first <= 64 channels");
    for (int i = 0; i < nk; ++i)
        if (cfg->kenc_channels[i] % 32) return fail(IMP_E_ARG, "keypoint_encoder channels must be multiples of 32");
    HIP_TRY(hipSetDevice(device));
    imp_ctx* c = new imp_ctx();
    c->cfg = *cfg;
    c->device = device;
    c->D = cfg->descriptor_dim;
    c->dh = c->D / IMP_NUM_HEADS;
    c->fuse_merge = 1; c->ot_compact = 0; c->ot_lane = 0;
    c->ot_resident = c->ot_local = c->ot_hier = c->ot_graph = 1; c->ot_verify = c->ot_fake = 0;
    c->use_wf = c->wf_chain = c->wf_fused = c->kv_image = 1; c->wf_fused_fake = 0;
    // the environment knows THREE switches of a context - the arithmetic, and the two kernels that wait for the whole chip (a process that shares the GPU turns them off) ...
    { const char* e = getenv("IMP_PRECISION"); c->prec = (e && !strcmp(e, "f32")) ? 0 : 1; }
    { const char* e = getenv("IMP_OT_RESIDENT"); if (e) c->ot_resident = atoi(e) != 0; }
    { const char* e = getenv("IMP_WF_FUSED"); if (e) c->wf_fused = atoi(e); }
    // ... everything else (step-down paths forced for A/B tests, fault-injection hooks) goes through imp_ctx_option; IMP_OPTIONS="name=value,..." applies a list at creation
    if (const char* e = getenv("IMP_OPTIONS")) {
        std::string list(e);
        size_t pos = 0;
        while (pos < list.size()) {
            size_t end = list.find(',', pos);
            if (end == std::string::npos) end = list.size();
            const std::string item = list.substr(pos, end - pos);
            const size_t eq = item.find('=');
            if (eq == std::string::npos || ctx_option(c, item.substr(0, eq).c_str(), strtol(item.c_str() + eq + 1, nullptr, 0)) != IMP_OK) {
                delete c;
                return fail(IMP_E_ARG, "IMP_OPTIONS: unknown option or missing value in '" + item + "'");
            }
            pos = end + 1;
        }
    }
    {   // CU count: sizes the resident Sinkhorn launches and the column-pass split of small weight-fragment GEMM launches
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) {
            delete c;
            return fail(IMP_E_HIP, "imp_create: cannot query the CU count of the device");
        }
        c->num_cus = cus;
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) != hipSuccess) xccs = 0;
        c->num_xccs = xccs;
    }
    {   // health words in mapped host memory: the kernels raise them, the entry points read them without synchronising
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 256, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            delete c;
            return fail(IMP_E_NOMEM, "imp_create: cannot allocate the mapped health words");
        }
        memset(h, 0, 256);                                 // 64 ints: [0] waiting-launch status, [1] range word, [2] graph tag wrap, [16..] post-mortem record (imp_kernels.h IMP_PM_BASE)
        c->xstatus_host = static_cast<int*>(h);
        c->xstatus_hostdev = static_cast<int*>(d);
        c->range_host = c->xstatus_host + 1;
        c->range_hostdev = c->xstatus_hostdev + 1;
    }
    c->kenc_maxc = c->D;
    for (int i = 0; i < nk; ++i) if (cfg->kenc_channels[i] > c->kenc_maxc) c->kenc_maxc = cfg->kenc_channels[i];
    build_schema(c);
    *out = c;
    return IMP_OK;
}

int imp_destroy(imp_ctx* c) {
    if (!c) return IMP_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_pool(c->allocs_w);
    free_pool(c->allocs_ws);
    free_pool(c->allocs_x);
    if (c->xstatus_host) (void)hipHostFree(c->xstatus_host);
    loop_release(c);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    delete c;
    return IMP_OK;
}

int imp_set_precision(imp_ctx* c, int precision) {
    if (!c || (precision != 0 && precision != 1)) return fail(IMP_E_ARG, "imp_set_precision: 0 (f32) or 1 (f16x3)");
    c->prec = precision;
    return IMP_OK;
}
int imp_get_precision(imp_ctx* c) { return c ? c->prec : -1; }
int imp_set_sinkhorn_storage(imp_ctx* c, int bytes) {
    if (!c || (bytes != 3 && bytes != 4)) return fail(IMP_E_ARG, "imp_set_sinkhorn_storage: 3 or 4 bytes per element");
    c->ot_compact = bytes == 3;
    return IMP_OK;
}

int imp_num_keys(imp_ctx* c) { return c ? (int)c->schema.size() : 0; }
const char* imp_key_name(imp_ctx* c, int i) {
    if (!c || i < 0 || i >= (int)c->schema.size()) return nullptr;
    return c->schema[i].c_str();
}

int imp_load_tensor(imp_ctx* c, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!c || !key || !data || ndim < 0 || ndim > 8) return fail(IMP_E_ARG, "imp_load_tensor: bad argument");
    bool known = false;
    for (const std::string& s : c->schema) if (s == key) { known = true; break; }
    if (!known) return fail(IMP_E_KEY, std::string("unexpected state_dict key: ") + key);
    int64_t numel = 1;
    HostTensor t;
    for (int i = 0; i < ndim; ++i) { numel *= shape[i]; t.shape.push_back(shape[i]); }
    t.data.assign(data, data + numel);
    c->raw[key] = std::move(t);
    c->finalized = false;
    return IMP_OK;
}

int imp_finalize_weights(imp_ctx* c) {
    if (!c) return fail(IMP_E_ARG, "null context");
    for (const std::string& s : c->schema)
        if (!c->raw.count(s)) return fail(IMP_E_KEY, "missing state_dict key: " + s);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    free_pool(c->allocs_w);
    const imp_config& cfg = c->cfg;
    const int D = c->D, dh = c->dh, nk = n_kenc(cfg);
    c->bin_score = c->raw["bin_score"].data[0];
    int rc;
    // keypoint encoder: [3, c1, ..., ck, D]
    c->kenc.assign(nk + 1, Linear());
    c->kenc_bn.assign(nk, NormC());
    int cin = 3;
    for (int i = 0; i <= nk; ++i) {
        const int cout = i < nk ? cfg.kenc_channels[i] : D;
        if ((rc = pack_linear(c, kname("kenc.encoder.%d", 3 * i), cout, cin, &c->kenc[i]))) return rc;
        if (i < nk && cfg.norm_fn == IMP_NORM_BN)
            if ((rc = pack_norm(c, kname("kenc.encoder.%d", 3 * i + 1), cout, &c->kenc_bn[i]))) return rc;
        cin = cout;
    }
    c->layers.assign(cfg.n_gnn_layers, GnnLayer());
    for (int li = 0; li < cfg.n_gnn_layers; ++li) {
        GnnLayer& L = c->layers[li];
        L.cross = cfg.layer_is_cross[li] != 0;
        L.shared = layer_shared(cfg, li);
        const std::string p = kname("gnn.layers.%d", li);
        const std::string pa = L.shared ? p : p + ".attn";
        // projections, output rows permuted to head-major
        const int nproj = L.shared ? 1 : 3;
        std::vector<float> W((size_t)nproj * D * D), b((size_t)nproj * D);
        for (int j = 0; j < nproj; ++j) {
            const std::string key = L.shared ? p + ".proj" : pa + kname(".proj.%d", j);
            const HostTensor* w = get(c, key + ".weight", (int64_t)D * D);
            const HostTensor* bb = get(c, key + ".bias", D);
            if (!w || !bb) return IMP_E_KEY;
            for (int r = 0; r < D; ++r) {
                const int src = ref_channel(r, dh);
                memcpy(&W[((size_t)j * D + r) * D], &w->data[(size_t)src * D], D * sizeof(float));
                b[(size_t)j * D + r] = bb->data[src];
            }
        }
        L.proj.out = nproj * D; L.proj.in = D;
        if ((rc = upload(c, &L.proj.W, W))) return rc;
        if ((rc = upload(c, &L.proj.b, b))) return rc;
        if ((rc = upload_wf(c, &L.proj_wf, W.data(), nproj * D, D))) return rc;
        // merge, input columns permuted to head-major
        {
            const HostTensor* w = get(c, pa + ".merge.weight", (int64_t)D * D);
            const HostTensor* bb = get(c, pa + ".merge.bias", D);
            if (!w || !bb) return IMP_E_KEY;
            std::vector<float> Wm((size_t)D * D);
            for (int o = 0; o < D; ++o)
                for (int k = 0; k < D; ++k) Wm[(size_t)o * D + k] = w->data[(size_t)o * D + ref_channel(k, dh)];
            L.merge.out = D; L.merge.in = D;
            if ((rc = upload(c, &L.merge.W, Wm))) return rc;
            if ((rc = upload(c, &L.merge.b, bb->data))) return rc;
        }
        if ((rc = pack_linear(c, p + ".mlp.0", 2 * D, 2 * D, &L.mlp0))) return rc;
        if (c->fuse_merge) {
            // mlp.0(cat[x, merge(o)]) = W1a x + (W1b Wm) o + (b1 + W1b bm): precomposed in fp64, rounded once to fp32
            const HostTensor* w1 = get(c, p + ".mlp.0.weight", (int64_t)4 * D * D);
            const HostTensor* b1 = get(c, p + ".mlp.0.bias", 2 * D);
            const HostTensor* wm = get(c, pa + ".merge.weight", (int64_t)D * D);
            const HostTensor* bm = get(c, pa + ".merge.bias", D);
            if (!w1 || !b1 || !wm || !bm) return IMP_E_KEY;
            std::vector<float> Wf((size_t)4 * D * D), bf(2 * D);
            std::vector<double> row(D);
            for (int o = 0; o < 2 * D; ++o) {
                const float* w1row = &w1->data[(size_t)o * 2 * D];
                memcpy(&Wf[(size_t)o * 2 * D], w1row, D * sizeof(float));
                for (int k = 0; k < D; ++k) row[k] = 0.0;
                double bacc = b1->data[o];
                for (int j = 0; j < D; ++j) {
                    const double a = w1row[D + j];
                    const float* wmrow = &wm->data[(size_t)j * D];
                    for (int k = 0; k < D; ++k) row[k] += a * (double)wmrow[k];
                    bacc += a * (double)bm->data[j];
                }
                for (int k = 0; k < D; ++k) Wf[(size_t)o * 2 * D + D + k] = (float)row[ref_channel(k, dh)];
                bf[o] = (float)bacc;
            }
            L.mlp0f.out = 2 * D; L.mlp0f.in = 2 * D;
            if ((rc = upload(c, &L.mlp0f.W, Wf))) return rc;
            if ((rc = upload(c, &L.mlp0f.b, bf))) return rc;
            if ((rc = upload_wf(c, &L.mlp0f_wf, Wf.data(), 2 * D, 2 * D))) return rc;
        }
        if (cfg.norm_fn == IMP_NORM_BN)
            if ((rc = pack_norm(c, p + ".mlp.1", 2 * D, &L.bn))) return rc;
        if ((rc = pack_linear(c, p + ".mlp.3", D, 2 * D, &L.mlp3))) return rc;
        {
            const HostTensor* w3 = get(c, p + ".mlp.3.weight", (int64_t)2 * D * D);
            if (!w3) return IMP_E_KEY;
            if ((rc = upload_wf(c, &L.mlp3_wf, w3->data.data(), D, 2 * D))) return rc;
        }
    }
    c->final_proj.assign(cfg.n_layers, Linear());
    for (int i = 0; i < cfg.n_layers; ++i)
        if ((rc = pack_linear(c, kname("final_proj.%d", i), D, D, &c->final_proj[i]))) return rc;
    c->finalized = true;
    c->cache[0].valid = c->cache[1].valid = false;
    if (cfg.max_batch > 0 && cfg.max_keypoints > 0) return ensure_workspace(c, cfg.max_batch, cfg.max_keypoints);
    return IMP_OK;
}

int imp_normalize_keypoints(imp_ctx* c, const float* kpts, int batch, int n, float width, float height, float* out,
                            void* stream) {
    if (!kpts || !out) return fail(IMP_E_ARG, "imp_normalize_keypoints: null argument");
    if (c) {
        HIP_TRY(hipSetDevice(c->device));
    } else {   // context-free use (the free function nets/layers.py:49-56): launch on the device that owns the buffer
        hipPointerAttribute_t at;
        HIP_TRY(hipPointerGetAttributes(&at, kpts));
        HIP_TRY(hipSetDevice(at.device));
    }
    HIP_TRY(launch_normalize_kpts(kpts, (long)batch * n, width, height, out, S(stream)));
    return IMP_OK;
}

int imp_encode_keypoints(imp_ctx* c, int batch, int n0, int n1, const float* nkpts0, const float* scores0,
                         const float* desc0, float* out0, const float* nkpts1, const float* scores1,
                         const float* desc1, float* out1, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!nkpts0 || !nkpts1 || !scores0 || !scores1 || !out0 || !out1) return fail(IMP_E_ARG, "imp_encode_keypoints: null argument");
    const int n[2] = {n0, n1};
    const float* kp[2] = {nkpts0, nkpts1};
    const float* sc[2] = {scores0, scores1};
    const float* de[2] = {desc0, desc1};
    float* out[2] = {out0, out1};
    return run_kenc(c, batch, n, kp, sc, 0.f, 0.f, de, out, S(stream));
}

int imp_forward_layer(imp_ctx* c, int layer_i, int batch, int n0, int n1, const float* desc0, const float* desc1,
                      float* out0, float* out1, const uint8_t* key_mask0, const uint8_t* key_mask1, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (layer_i < 0 || layer_i >= c->cfg.n_gnn_layers) return fail(IMP_E_ARG, "layer index out of range");
    if (!desc0 || !desc1 || !out0 || !out1) return fail(IMP_E_ARG, "imp_forward_layer: null argument");
    const int n[2] = {n0, n1};
    const float* de[2] = {desc0, desc1};
    float* out[2] = {out0, out1};
    const uint8_t* km[2] = {key_mask0, key_mask1};
    return run_layer(c, layer_i, batch, n, de, out, km, S(stream));
}

int imp_attention_prob(imp_ctx* c, int which, float* prob, void* stream) {
    if (c && c->rc.on) return fail(IMP_E_ARG, "imp_attention_prob: not with per-pair keypoint counts (imp_set_counts) - the cached attention of a ragged batch has per-pair shapes");
    if (!c || !prob || which < 0 || which > 3) return fail(IMP_E_ARG, "imp_attention_prob: bad argument");
    const ProbSpec ps = kProb[which];
    const AttnCache& cache = c->cache[ps.kind];
    if (!cache.valid) return fail(IMP_E_STATE, "no cached attention of that kind");
    HIP_TRY(hipSetDevice(c->device));
    const int D = c->D, dh = c->dh, nq = cache.n[ps.qside], nk = cache.n[ps.kside];
    GemmParams p = gemm_defaults(c, dh);
    p.nside = 1; p.nsub = IMP_NUM_HEADS;
    GemmSide& g = p.side[0];
    long ldk = 0;
    hipError_t kerr;
    const float* kf = cached_k_fp32(c, ps.kind, ps.kside, &ldk, S(stream), &kerr);
    HIP_TRY(kerr);
    g.A = c->qkv[ps.kind][ps.qside]; g.W = kf; g.C = prob;
    g.rowvec = c->lse[ps.kind][ps.qside];
    g.M = nq; g.N = nk;
    g.sA_b = (long)nq * 3 * D; g.sA_s = dh; g.sW_b = (long)nk * ldk; g.sW_s = dh;
    g.sC_b = (long)IMP_NUM_HEADS * nq * nk; g.sC_s = (long)nq * nk;
    g.sRV_b = (long)IMP_NUM_HEADS * nq; g.sRV_s = nq;
    p.lda = 3 * D; p.ldw = (int)ldk; p.ldc = nk;
    p.flags = GEMM_EPI_DIV | GEMM_EPI_EXPROW;
    p.div = (float)std::sqrt((double)dh);
    HIP_TRY(launch_gemm_f32(p, cache.batch, S(stream)));
    if (cache.masked[ps.kside]) {
        const long total = (long)cache.batch * IMP_NUM_HEADS * nq * nk;
        hipLaunchKernelGGL(zero_masked_columns_kernel, dim3((total + 255) / 256), dim3(256), 0, S(stream), prob,
                           c->cmask[ps.kind][ps.kside], nq, nk, total);
        HIP_TRY(hipGetLastError());
    }
    return IMP_OK;
}

int imp_attention_received(imp_ctx* c, int which, float* out, void* stream) {
    if (c && c->rc.on) return fail(IMP_E_ARG, "imp_attention_received: not with per-pair keypoint counts (imp_set_counts) - the cached attention of a ragged batch has per-pair shapes");
    if (!c || !out || which < 0 || which > 3) return fail(IMP_E_ARG, "imp_attention_received: bad argument");
    const ProbSpec ps = kProb[which];
    const AttnCache& cache = c->cache[ps.kind];
    if (!cache.valid) return fail(IMP_E_STATE, "no cached attention of that kind");
    HIP_TRY(hipSetDevice(c->device));
    const int D = c->D, nq = cache.n[ps.qside], nk = cache.n[ps.kside];
    ColsumParams p;
    memset(&p, 0, sizeof p);
    long ldk = 0;
    hipError_t kerr;
    const float* kf = cached_k_fp32(c, ps.kind, ps.kside, &ldk, S(stream), &kerr);
    HIP_TRY(kerr);
    p.nside = 1; p.ldq = 3 * D; p.ldk = (int)ldk; p.dh = c->dh;
    ColsumSide& g = p.side[0];
    g.q = c->qkv[ps.kind][ps.qside]; g.k = kf; g.lse = c->lse[ps.kind][ps.qside];
    g.out = c->colsum[which];
    g.kmask = cache.masked[ps.kside] ? c->cmask[ps.kind][ps.kside] : nullptr;
    g.nq = nq; g.nk = nk; g.sq_b = (long)nq * 3 * D; g.sk_b = (long)nk * ldk;
    HIP_TRY(launch_attn_colsum(p, cache.batch, c->prec, S(stream)));
    for (int b = 0; b < cache.batch; ++b)
        HIP_TRY(launch_attn_mass_normalize(c->colsum[which] + (size_t)b * IMP_NUM_HEADS * nk, nk, out + (size_t)b * nk,
                                           S(stream)));
    return IMP_OK;
}

int imp_score_mass(imp_ctx* c, int n0, int n1, const float* scores, float* mass0, float* mass1, void* stream) {
    int rc = check_ready(c, 1, n0, n1);
    if (rc) return rc;
    if (!scores || !mass0 || !mass1) return fail(IMP_E_ARG, "imp_score_mass: null argument");
    HIP_TRY(launch_score_mass(scores, n0, n1, mass0, mass1, c->colpart_v, S(stream)));
    return IMP_OK;
}

int imp_pool_select(imp_ctx* c, int n, const float* mass, const float* a_self, const float* a_cross, float thr,
                    int64_t* ids, int32_t* counts, void* stream) {
    if (!c || !mass || !a_self || !a_cross || !ids || !counts || n < 1) return fail(IMP_E_ARG, "imp_pool_select: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    PoolSide ps[2];
    ps[0] = PoolSide{mass, a_self, a_cross, ids, n, 0};
    ps[1] = ps[0];
    HIP_TRY(launch_pool_select(ps, 1, thr, counts, S(stream)));
    return IMP_OK;
}

int imp_pool_select_pair(imp_ctx* c, int n0, const float* mass0, const float* a_self0, const float* a_cross0, int skip0, int64_t* ids0,
                         int n1, const float* mass1, const float* a_self1, const float* a_cross1, int skip1, int64_t* ids1, float thr,
                         int32_t* counts, void* stream) {
    if (!c || !mass0 || !a_self0 || !a_cross0 || !ids0 || !mass1 || !a_self1 || !a_cross1 || !ids1 || !counts || n0 < 1 || n1 < 1)
        return fail(IMP_E_ARG, "imp_pool_select_pair: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    PoolSide ps[2];
    ps[0] = PoolSide{mass0, a_self0, a_cross0, ids0, n0, skip0 ? 1 : 0};
    ps[1] = PoolSide{mass1, a_self1, a_cross1, ids1, n1, skip1 ? 1 : 0};
    HIP_TRY(launch_pool_select(ps, 2, thr, counts, S(stream)));
    return IMP_OK;
}

int imp_compute_distance(imp_ctx* c, int layer_id, int batch, int n0, int n1, const float* desc0, const float* desc1,
                         float* dist, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!desc0 || !desc1 || !dist) return fail(IMP_E_ARG, "imp_compute_distance: null argument");
    const int n[2] = {n0, n1};
    const float* de[2] = {desc0, desc1};
    return run_distance(c, layer_id, batch, n, de, dist, S(stream));
}

int imp_compute_score(imp_ctx* c, int batch, int n0, int n1, const float* dist, float bin_score, int iterations,
                      int with_sinkhorn, float* scores, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!dist) return fail(IMP_E_ARG, "imp_compute_score: null dist");
    return run_score(c, batch, n0, n1, dist, bin_score, iterations, with_sinkhorn, scores, nullptr, S(stream));
}

int imp_compute_matches(imp_ctx* c, int batch, int n0, int n1, const float* scores, float p, int64_t* indices0,
                        int64_t* indices1, float* mscores0, float* mscores1, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!scores) return fail(IMP_E_ARG, "imp_compute_matches: null scores");
    if (c->rc.on) return fail(IMP_E_ARG, "imp_compute_matches: no score tensors with per-pair keypoint counts (imp_set_counts): use imp_match_tail");
    HIP_TRY(launch_score_maxima(scores, batch, n0, n1, c->max0, c->arg0, c->max1, c->arg1, c->colpart_v, c->colpart_i,
                                S(stream)));
    HIP_TRY(launch_mutual_matches(batch, n0, n1, c->max0, c->arg0, c->max1, c->arg1, p, indices0, indices1, mscores0,
                                  mscores1, c->range_hostdev, S(stream)));
    return IMP_OK;
}

// AdaGMN.pool for pair b of the cached batch: the pair's own sizes m[0], m[1] inside a batch whose tensors are padded to pad[0], pad[1]
// (b = 0 and m == pad: the single pair imp_pool has always served).  Every kernel runs on the pair's slice with the pair's sizes, so
// the result is the one of the pair pooled alone.
static int pool_pair_impl(imp_ctx* c, int b, const int m[2], const int pad[2], const float* scores, float thr, int n_min_tokens,
                          int64_t* ids0, int64_t* ids1, int32_t* counts, hipStream_t st) {
    const int D = c->D;
    // attention mass received per key: a00, a11 (self) and a01 (img1 queries -> img0 keys), a10 (nets/adgm.py:557-565)
    for (int kind = 0; kind < 2; ++kind) {
        ColsumParams p;
        memset(&p, 0, sizeof p);
        p.nside = 2; p.ldq = 3 * D; p.dh = c->dh;
        long ldk = 3 * D;
        for (int keyside = 0; keyside < 2; ++keyside) {
            const int qside = kind == 0 ? keyside : 1 - keyside;
            ColsumSide& g = p.side[keyside];
            hipError_t kerr;
            const float* kf = cached_k_fp32(c, kind, keyside, &ldk, st, &kerr);       // (the whole batch: one small launch per call)
            HIP_TRY(kerr);
            g.q = c->qkv[kind][qside] + (size_t)b * pad[qside] * 3 * D;
            g.k = kf + (size_t)b * pad[keyside] * ldk;
            g.lse = c->lse[kind][qside] + (size_t)b * IMP_NUM_HEADS * pad[qside];
            g.lq = pad[qside];
            p.ldk = (int)ldk;
            g.out = c->colsum[kind * 2 + keyside];
            g.kmask = c->cache[kind].masked[keyside] ? c->cmask[kind][keyside] + (size_t)b * pad[keyside] : nullptr;   // masked keys received exactly 0
            g.nq = m[qside];
            g.nk = m[keyside];
            g.sq_b = 0; g.sk_b = 0;
        }
        HIP_TRY(launch_attn_colsum(p, 1, c->prec, st));
        for (int keyside = 0; keyside < 2; ++keyside)
            HIP_TRY(launch_attn_mass_normalize(c->colsum[kind * 2 + keyside], m[keyside], c->amass[kind * 2 + keyside], st));
    }
    HIP_TRY(launch_score_mass(scores, m[0], m[1], c->mass[0], c->mass[1], c->colpart_v, st));
    PoolSide ps[2];
    // image 0: a_self = a00 (colsum[0]), a_cross = a01 (colsum[2]); image 1: a_self = a11 (colsum[1]), a_cross = a10 (colsum[3])
    ps[0] = PoolSide{c->mass[0], c->amass[0], c->amass[2], ids0, m[0], (n_min_tokens > 0 && m[0] + 1 <= n_min_tokens) ? 1 : 0};
    ps[1] = PoolSide{c->mass[1], c->amass[1], c->amass[3], ids1, m[1], (n_min_tokens > 0 && m[1] + 1 <= n_min_tokens) ? 1 : 0};
    HIP_TRY(launch_pool_select(ps, 2, thr, counts, st));
    return IMP_OK;
}

int imp_pool(imp_ctx* c, int n0, int n1, const float* scores, float mscore_th, float uncertainty_ratio, int n_min_tokens,
             int64_t* ids0, int64_t* ids1, int32_t* counts, void* stream) {
    int rc = check_ready(c, 1, n0, n1);
    if (rc) return rc;
    if (!scores || !ids0 || !ids1 || !counts) return fail(IMP_E_ARG, "imp_pool: null argument");
    if (c->rc.on) return fail(IMP_E_ARG, "imp_pool: not with per-pair keypoint counts (imp_set_counts): imp_pool_pair");
    for (int k = 0; k < 2; ++k)
        if (!c->cache[k].valid || c->cache[k].n[0] != n0 || c->cache[k].n[1] != n1)
            return fail(IMP_E_STATE, "imp_pool: cached self/cross attention does not match (n0, n1)");
    const int m[2] = {n0, n1};
    return pool_pair_impl(c, 0, m, m, scores, mscore_th * uncertainty_ratio, n_min_tokens, ids0, ids1, counts, S(stream));
}

int imp_pool_pair(imp_ctx* c, int pair, int batch, int n0, int n1, const float* scores, float mscore_th, float uncertainty_ratio,
                  int n_min_tokens, int64_t* ids0, int64_t* ids1, int32_t* counts, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!scores || !ids0 || !ids1 || !counts || pair < 0 || pair >= batch) return fail(IMP_E_ARG, "imp_pool_pair: bad argument");
    for (int k = 0; k < 2; ++k)
        if (!c->cache[k].valid || c->cache[k].batch != batch || c->cache[k].n[0] != n0 || c->cache[k].n[1] != n1)
            return fail(IMP_E_STATE, "imp_pool_pair: cached self/cross attention does not match (batch, n0, n1)");
    const int pad[2] = {n0, n1};
    const int m[2] = {c->rc.on ? c->rc.n[0][pair] : n0, c->rc.on ? c->rc.n[1][pair] : n1};
    if (m[0] < 1 || m[1] < 1) return fail(IMP_E_ARG, "imp_pool_pair: the pair is retired (count 0)");
    return pool_pair_impl(c, pair, m, pad, scores, mscore_th * uncertainty_ratio, n_min_tokens, ids0, ids1, counts, S(stream));
}

int imp_gather_rows(imp_ctx* c, int batch, int n_in, int n_out, int dim, const float* in, const int64_t* ids, float* out,
                    void* stream) {
    if (!c || !in || !ids || !out || dim % 4) return fail(IMP_E_ARG, "imp_gather_rows: bad argument (dim % 4 == 0)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_gather_rows(in, ids, out, batch, n_in, n_out, dim, S(stream)));
    return IMP_OK;
}

int imp_masked_commit(imp_ctx* c, int n0sel, const int64_t* gids0, const int64_t* gids1, const int64_t* indices0, const float* mscores0,
                      int64_t* out_indices0, float* out_mscores0, const int64_t* keep0, int nkeep0, const int64_t* keep1, int nkeep1,
                      int64_t* new_gids0, int64_t* new_gids1, uint8_t* mask0, uint8_t* mask1, void* stream) {
    if (!c || n0sel < 0 || !gids0 || !gids1 || !indices0 || !mscores0 || !out_indices0 || !out_mscores0)
        return fail(IMP_E_ARG, "imp_masked_commit: bad argument");
    if ((new_gids0 != nullptr) != (new_gids1 != nullptr) || (new_gids0 && (!mask0 || !mask1 || nkeep0 < 0 || nkeep1 < 0)))
        return fail(IMP_E_ARG, "imp_masked_commit: the id-list update needs both lists and both mask rows");
    HIP_TRY(hipSetDevice(c->device));
    MaskedCommit p{gids0, gids1, indices0, mscores0, out_indices0, out_mscores0, n0sel, keep0, keep1, nkeep0, nkeep1, new_gids0, new_gids1, mask0, mask1};
    HIP_TRY(launch_masked_commit(p, S(stream)));
    return IMP_OK;
}

// In-call recovery (imp_set_range_recovery; VERDICT r4 #5b, r5 #2a): after a call's work is enqueued the entry point WAITS for it (one host
// synchronisation per call) and looks at the two words the kernels raise in mapped host memory:
//   * the range word (a match kernel met non-finite scores: an operand left the fp16 range of the f16x3 arithmetic): `run` is executed again on the
//     native fp32 MFMA path (c->prec = 0), which has no operand limit;
//   * the waiting-launch word (round 6: a chip-resident Sinkhorn or a fused layer launch of THIS call timed out in an exchange and voided its outputs):
//     the context takes its usual step down (resident_health: chip-wide exchange / streaming kernels / two-launch layers) and `run` is executed
//     again - the caller never sees the void answer (the reference's calls always answer: nets/gm.py:145-247).  Whether the voided launch is this
//     call's is read from its tag in the post-mortem record; an OLDER launch's time-out (a step call nobody waited for) is reported as before:
//     IMP_E_RESIDENT, that earlier call is void.
// Returns IMP_OK (nothing happened, or recovered: the outputs hold the repaired results) or an error.
static int run_with_recovery(imp_ctx* c, hipStream_t st, const std::function<int()>& run) {
    const unsigned sk0 = c->xtag, fx0 = c->fx_tag;         // tags this call's waiting launches will draw from
    int rc = run();
    if (rc) return rc;
    if (!c->range_recover || !c->range_host) return IMP_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return IMP_OK;      // a capture cannot wait
    // (a polled wait - hipStreamQuery for up to 20 ms before blocking - was measured and is no faster: configs[1] 1.68-1.70 vs 1.65-1.71 ms per call; the
    // price of the wait is the host's enqueue work no longer running ahead of the GPU, not the wake-up; profiles/r05/c2_latency_range_recovery_ab.log)
    for (int attempt = 0; attempt < 4; ++attempt) {
        HIP_TRY(hipStreamSynchronize(st));
        const int wst = c->xstatus_host ? *static_cast<volatile int*>(c->xstatus_host) : 0;
        if (wst) {
            const unsigned vt = (unsigned)static_cast<volatile int*>(c->xstatus_host)[IMP_PM_BASE + 1];
            const int kind = static_cast<volatile int*>(c->xstatus_host)[IMP_PM_BASE];
            const bool ours = kind == 3 ? (vt > fx0 && vt <= c->fx_tag) : (vt >= sk0 && vt < c->xtag && c->xtag >= sk0);
            const int hrc = resident_health(c);             // waits, resets the exchange state, steps the context down; IMP_E_RESIDENT
            if (!ours || hrc != IMP_E_RESIDENT || attempt == 3) return hrc;
            c->resident_repaired += 1;
            if ((rc = run())) return rc;
            continue;
        }
        if (c->prec != 1 || !*static_cast<volatile int*>(c->range_host)) return IMP_OK;
        *static_cast<volatile int*>(c->range_host) = 0;
        c->range_events += 1;
        c->prec = 0;
        rc = run();
        c->prec = 1;
        if (rc) return rc;
        c->range_recovered += 1;
        // (the fp32 pass may itself meet a voided waiting launch: look again)
    }
    return IMP_OK;
}

static int match_pair_enqueue(imp_ctx* c, int batch, int n0, int n1, const float* kpts0, const float* scores0, const float* desc0,
                              const float* kpts1, const float* scores1, const float* desc1, float width, float height,
                              float bin_score, int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0,
                              float* mscores0, int64_t* indices1, float* mscores1, float* scores, hipStream_t st);

int imp_match_pair(imp_ctx* c, int batch, int n0, int n1, const float* kpts0, const float* scores0, const float* desc0,
                   const float* kpts1, const float* scores1, const float* desc1, float width, float height,
                   float bin_score, int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0,
                   float* mscores0, int64_t* indices1, float* mscores1, float* scores, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!kpts0 || !kpts1 || !scores0 || !scores1 || !desc0 || !desc1) return fail(IMP_E_ARG, "imp_match_pair: null input");
    if (c->rc.on && scores) return fail(IMP_E_ARG, "imp_match_pair: no score tensor for a ragged batch (imp_set_counts): every pair's dustbin row / column sits elsewhere");
    hipStream_t st = S(stream);
    auto run = [&]() {
        return match_pair_enqueue(c, batch, n0, n1, kpts0, scores0, desc0, kpts1, scores1, desc1, width, height, bin_score, sinkhorn_iterations,
                                  with_sinkhorn, p, indices0, mscores0, indices1, mscores1, scores, st);
    };
    return run_with_recovery(c, st, run);
}

static int match_pair_enqueue(imp_ctx* c, int batch, int n0, int n1, const float* kpts0, const float* scores0, const float* desc0,
                              const float* kpts1, const float* scores1, const float* desc1, float width, float height,
                              float bin_score, int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0,
                              float* mscores0, int64_t* indices1, float* mscores1, float* scores, hipStream_t st) {
    int rc = IMP_OK;
    const int n[2] = {n0, n1};
    const float* kp[2] = {kpts0, kpts1};
    const float* sc[2] = {scores0, scores1};
    const float* de[2] = {desc0, desc1};
    float* dw[2] = {c->descw[0], c->descw[1]};
    // host-side section timers (a development build flips this: microseconds of enqueue work per section, printed every 50 calls)
    constexpr bool hprof = false;
    static double hp_acc[5] = {0, 0, 0, 0, 0}; static int hp_n = 0;
    auto hp_now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double hp_t = hprof ? hp_now() : 0.0;
    auto hp_mark = [&](int i) { if (hprof) { const double t = hp_now(); hp_acc[i] += t - hp_t; hp_t = t; } };
    if ((rc = run_kenc(c, batch, n, kp, sc, width, height, de, dw, st))) return rc;   // desc + enc (nets/gm.py:177-178)
    hp_mark(0);
    const uint8_t* nomask[2] = {nullptr, nullptr};
    const float* dr[2] = {c->descw[0], c->descw[1]};
    bool proj_done = false;                                // inside this call the layers chain: layer i's last launch also projects for layer i + 1
    for (int li = 0; li < c->cfg.n_gnn_layers && !rc; ++li) {
        bool chained = false;
        rc = run_layer(c, li, batch, n, dr, dw, nomask, st, proj_done, li + 1 < c->cfg.n_gnn_layers ? li + 1 : -1, &chained);
        proj_done = chained;
    }
    if (rc) return rc;
    hp_mark(1);
    if ((rc = run_distance(c, c->cfg.n_layers - 1, batch, n, dr, c->dist, st))) return rc;
    hp_mark(2);
    OtBuffers o;
    bool max_done = false;
    if ((rc = run_score(c, batch, n0, n1, c->dist, bin_score, sinkhorn_iterations, with_sinkhorn, scores, &o, st, &max_done))) return rc;
    hp_mark(3);
    if (hprof && ++hp_n % 50 == 0) {
        fprintf(stderr, "[host prof] enqueue us per call: encoder %.0f  layers %.0f  distance %.0f  score %.0f\n", hp_acc[0] / 50, hp_acc[1] / 50, hp_acc[2] / 50, hp_acc[3] / 50);
        for (double& a : hp_acc) a = 0;
    }
    if (!max_done) HIP_TRY(launch_ot_maxima(batch, n0, n1, with_sinkhorn ? 0 : 1, o, c->max0, c->arg0, c->max1, c->arg1, st));
    HIP_TRY(launch_mutual_matches(batch, n0, n1, c->max0, c->arg0, c->max1, c->arg1, p, indices0, indices1, mscores0,
                                  mscores1, c->range_hostdev, st, &c->rc));
    return IMP_OK;
}

int imp_match_tail(imp_ctx* c, int layer_id, int batch, int n0, int n1, const float* desc0, const float* desc1, float bin_score,
                   int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0, float* mscores0, int64_t* indices1,
                   float* mscores1, void* stream) {
    return imp_match_tail_scores(c, layer_id, batch, n0, n1, desc0, desc1, bin_score, sinkhorn_iterations, with_sinkhorn, p, indices0, mscores0,
                                 indices1, mscores1, nullptr, stream);
}

int imp_match_tail_scores(imp_ctx* c, int layer_id, int batch, int n0, int n1, const float* desc0, const float* desc1, float bin_score,
                          int sinkhorn_iterations, int with_sinkhorn, float p, int64_t* indices0, float* mscores0, int64_t* indices1,
                          float* mscores1, float* scores, void* stream) {
    int rc = check_ready(c, batch, n0, n1);
    if (rc) return rc;
    if (!desc0 || !desc1) return fail(IMP_E_ARG, "imp_match_tail: null descriptors");
    if (scores && !with_sinkhorn) return fail(IMP_E_ARG, "imp_match_tail_scores: the score tensor comes with the Sinkhorn scorer only");
    hipStream_t st = S(stream);
    const int n[2] = {n0, n1};
    const float* de[2] = {desc0, desc1};
    auto run = [&]() -> int {
        int rc2;
        if ((rc2 = run_distance(c, layer_id, batch, n, de, c->dist, st))) return rc2;
        OtBuffers o;
        bool max_done = false;
        if ((rc2 = run_score(c, batch, n0, n1, c->dist, bin_score, sinkhorn_iterations, with_sinkhorn, scores, &o, st, &max_done))) return rc2;
        if (!max_done) HIP_TRY(launch_ot_maxima(batch, n0, n1, with_sinkhorn ? 0 : 1, o, c->max0, c->arg0, c->max1, c->arg1, st));
        HIP_TRY(launch_mutual_matches(batch, n0, n1, c->max0, c->arg0, c->max1, c->arg1, p, indices0, indices1, mscores0,
                                      mscores1, c->range_hostdev, st, &c->rc));
        return IMP_OK;
    };
    // (recovery mode: an overflow or a voided resident launch INSIDE the tail - final projection, distance, Sinkhorn - is repaired here; descriptors that
    // arrive non-finite from the caller's earlier layer calls stay void and are reported at the next entry point, as without the mode)
    return run_with_recovery(c, st, run);
}

int imp_loop_lockstep(imp_ctx* c, int B, const int32_t* n0v, const int32_t* n1v, int n0, int n1, const float* nkpts0, const float* scores0,
                      const float* desc0, const float* nkpts1, const float* scores1, const float* desc1, float bin_score, int sinkhorn_iterations,
                      int n_iterations, unsigned valid_mask, float match_ratio, int min_kpts, double error_th, double stop_pose_deg,
                      int pose_threads, int pose_iterations, unsigned pose_seed, int pose_flags, imp_loop_pair* pairs, void* stream) {
    if (!c || !n0v || !n1v || !pairs || B < 1 || B > IMP_RAGGED_MAX) return fail(IMP_E_ARG, "imp_loop_lockstep: 1 .. 16 pairs, counts and pair records");
    if (!nkpts0 || !nkpts1 || !scores0 || !scores1 || !desc0 || !desc1) return fail(IMP_E_ARG, "imp_loop_lockstep: null input");
    if (match_ratio > 0.2f) return fail(IMP_E_ARG, "imp_loop_lockstep: match_ratio must be <= 0.2 (the final p = 0.2 matches are derived from the scored ones)");
    if (n_iterations < 1 || 2 * n_iterations > c->cfg.n_gnn_layers) return fail(IMP_E_ARG, "imp_loop_lockstep: more iterations than the model has layer pairs");
    int rc = imp_set_counts(c, B, n0v, n1v);
    if (rc) return rc;
    struct Restore { imp_ctx* c; ~Restore() { c->rc.on = 0; c->rc_batch = 0; } } restore{c};
    if ((rc = check_ready(c, B, n0, n1))) return rc;
    hipStream_t st = S(stream);
    const int n[2] = {n0, n1};
    // buffers of a scored iteration: device outputs + one pinned mirror (int64 indices | float scores)
    const size_t need = (size_t)B * n0;
    if (need > c->lp_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c->lp_idx) (void)hipFree(c->lp_idx);
        if (c->lp_ms) (void)hipFree(c->lp_ms);
        if (c->lp_pin) (void)hipHostFree(c->lp_pin);
        c->lp_idx = nullptr; c->lp_ms = nullptr; c->lp_pin = nullptr; c->lp_cap = 0;
        const size_t cap = need < 16384 ? 16384 : need;
        HIP_TRY(hipMalloc(&c->lp_idx, cap * sizeof(int64_t)));
        HIP_TRY(hipMalloc(&c->lp_ms, cap * sizeof(float)));
        HIP_TRY(hipHostMalloc(&c->lp_pin, cap * 12));
        c->lp_cap = cap;
    }
    if (!c->lp_ev) HIP_TRY(hipEventCreateWithFlags(&c->lp_ev, hipEventDisableTiming));
    const bool with_pose = pose_threads > 0;
    if (with_pose && !c->lp_workers) c->lp_workers = new PoseWorkers(pose_threads < B ? B : pose_threads, c->device);
    int64_t* const h_idx = reinterpret_cast<int64_t*>(c->lp_pin);
    float* const h_ms = reinterpret_cast<float*>(c->lp_pin + c->lp_cap * sizeof(int64_t));
    // When does a pair's exit test see its pose estimate?  Deferred (rounds 4a: the estimate of scored iteration k runs beside the next two
    // iterations' layers and decides at the next scored iteration - right when an estimate costs more than two iterations of layers) or
    // immediately (the group waits for this iteration's estimates, which run side by side: nothing is computed for a pair after its exit).
    // The results are the same; immediate is the faster one on the harder set (profiles/r05/MEASURED.md) and the only one kept reachable.
    const bool immediate = with_pose;

    std::vector<LoopPair> P(B);
    struct WaitAll {                                    // no return path may leave a pose worker behind that still reads the caller's arrays
        std::vector<LoopPair>& P;
        ~WaitAll() { for (auto& q : P) if (q.pend) q.pend->fut.wait(); }
    } wait_all{P};
    for (int b = 0; b < B; ++b) { pairs[b].found = 0; pairs[b].n_iterations = n_iterations; }
    auto retire = [&](int b) { P[b].live = false; c->rc.n[0][b] = 0; c->rc.n[1][b] = 0; };
    // finish pair b's outstanding pose estimate and take its exit test (eval/matching.py:84-117); true when the pair exits
    auto resolve = [&](int b) -> bool {
        LoopPair& q = P[b];
        if (q.pend_it < 0) return false;
        const int it_k = q.pend_it;
        q.pend_it = -1;
        std::shared_ptr<LoopPose> job = q.pend;
        q.pend.reset();
        bool have = false;
        if (job) { job->fut.wait(); have = job->rc == 0; }
        double diff_R = INFINITY, diff_t = INFINITY;
        if (it_k >= 1 && have && q.has_last) { diff_R = loop_angle_mat(q.lastR, job->R); diff_t = loop_angle_vec(q.lastT, job->t); }
        q.has_last = have;
        if (have) { memcpy(q.lastR, job->R, sizeof q.lastR); memcpy(q.lastT, job->t, sizeof q.lastT); }
        const double pose_diff = diff_R > diff_t ? diff_R : diff_t;
        if (stop_pose_deg >= 0.0 && pose_diff <= stop_pose_deg) {                  // eval/matching.py:110-117
            imp_loop_pair& o = pairs[b];
            const int nb = n0v[b];
            for (int i = 0; i < nb; ++i) { o.indices0[i] = -1; o.mscores0[i] = q.pend_ms[i]; }
            for (size_t m = 0; m < q.pm0.size(); ++m)
                if (job->mask[m]) o.indices0[q.pm0[m]] = q.pm1[m];
            memcpy(o.R, job->R, sizeof o.R); memcpy(o.t, job->t, sizeof o.t);
            o.found = 1; o.n_iterations = it_k + 1;
            retire(b);
            return true;
        }
        return false;
    };

    const float* kp[2] = {nkpts0, nkpts1};
    const float* sc[2] = {scores0, scores1};
    const float* de[2] = {desc0, desc1};
    float* dw[2] = {c->descw[0], c->descw[1]};
    if ((rc = run_kenc(c, B, n, kp, sc, 0.f, 0.f, de, dw, st))) return rc;       // desc + enc (eval/matching.py:47-50)
    const uint8_t* nomask[2] = {nullptr, nullptr};
    const float* dr[2] = {c->descw[0], c->descw[1]};
    bool proj_done = false;
    int layers_done = -1;
    auto run_two_layers = [&](int it) -> int {
        for (int li = 2 * it; li <= 2 * it + 1; ++li) {
            bool chained = false;
            const int r2 = run_layer(c, li, B, n, dr, dw, nomask, st, proj_done, li + 1 < c->cfg.n_gnn_layers ? li + 1 : -1, &chained);
            if (r2) return r2;
            proj_done = chained;
        }
        layers_done = it;
        return IMP_OK;
    };
    for (int it = 0; it < n_iterations; ++it) {
        if (layers_done < it && (rc = run_two_layers(it))) return rc;
        if (!((valid_mask >> it) & 1u)) continue;
        // score + matches of this iteration for the whole (ragged) batch, one copy, and - before the host looks at it - the next
        // iteration's layers, so that the GPU works through the pose estimates
        if ((rc = run_distance(c, it, B, n, dr, c->dist, st))) return rc;
        OtBuffers o;
        bool max_done = false;
        if ((rc = run_score(c, B, n0, n1, c->dist, bin_score, sinkhorn_iterations, 1, nullptr, &o, st, &max_done))) return rc;
        if (!max_done) HIP_TRY(launch_ot_maxima(B, n0, n1, 0, o, c->max0, c->arg0, c->max1, c->arg1, st));      // (a one-pair group on the streaming kernels)
        HIP_TRY(launch_mutual_matches(B, n0, n1, c->max0, c->arg0, c->max1, c->arg1, match_ratio, c->lp_idx, nullptr, c->lp_ms, nullptr,
                                      c->range_hostdev, st, &c->rc));
        HIP_TRY(hipMemcpyAsync(h_idx, c->lp_idx, need * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_ms, c->lp_ms, need * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c->lp_ev, st));
        if (!immediate && it + 1 < n_iterations && (rc = run_two_layers(it + 1))) return rc;
        HIP_TRY(hipEventSynchronize(c->lp_ev));
        if ((rc = resident_health(c))) return rc;                                  // a voided launch: the caller runs the group again
        bool retired = false;
        for (int b = 0; b < B; ++b) {
            LoopPair& q = P[b];
            if (!q.live) continue;
            if (resolve(b)) { retired = true; continue; }
            const int nb = n0v[b];
            const int64_t* ib = h_idx + (size_t)b * n0;
            const float* mb = h_ms + (size_t)b * n0;
            q.last_idx.assign(ib, ib + nb);
            q.last_ms.assign(mb, mb + nb);
            q.scored = true;
            q.pm0.clear(); q.pm1.clear();
            for (int i = 0; i < nb; ++i)
                if (ib[i] > -1) { q.pm0.push_back(i); q.pm1.push_back((int)ib[i]); }
            if ((int)q.pm0.size() < min_kpts) { q.has_last = false; continue; }   // eval/matching.py:63-66
            q.pend_it = it;
            q.pend_idx = q.last_idx; q.pend_ms = q.last_ms;
            q.pend.reset();
            if (with_pose) {
                auto job = std::make_shared<LoopPose>();
                const int m = (int)q.pm0.size();
                job->n = m;
                job->k0.resize(2 * (size_t)m); job->k1.resize(2 * (size_t)m); job->mask.assign(m, 0);
                for (int j = 0; j < m; ++j) {
                    job->k0[2 * j] = pairs[b].pts0[2 * q.pm0[j]]; job->k0[2 * j + 1] = pairs[b].pts0[2 * q.pm0[j] + 1];
                    job->k1[2 * j] = pairs[b].pts1[2 * q.pm1[j]]; job->k1[2 * j + 1] = pairs[b].pts1[2 * q.pm1[j] + 1];
                }
                job->fut = job->done.get_future();
                const double* K0 = pairs[b].K0; const double* K1 = pairs[b].K1;
                const int dev = c->device;
                c->lp_workers->submit([job, K0, K1, error_th, pose_iterations, pose_seed, pose_flags, dev](hipStream_t ps) {
                    job->rc = imp_estimate_pose(job->k0.data(), job->k1.data(), job->n, K0, K1, error_th, pose_iterations, pose_seed, dev, job->E, job->R,
                                                job->t, job->mask.data(), nullptr, &job->ninl, pose_flags, ps);
                    job->done.set_value();
                });
                q.pend = job;
            }
        }
        if (immediate)                                                             // this iteration's estimates decide now (they ran side by side)
            for (int b = 0; b < B; ++b)
                if (P[b].live && resolve(b)) retired = true;
        bool any = false;
        for (int b = 0; b < B; ++b) any = any || P[b].live;
        if (!any) break;
        (void)retired;                                                             // (the counts live in c->rc: the next launches see them)
    }
    for (int b = 0; b < B; ++b)
        if (P[b].live) resolve(b);                                                 // the estimates of the last scored iteration
    // wait for estimates nobody looked at any more (a worker must not outlive its job's buffers)
    for (int b = 0; b < B; ++b)
        if (P[b].pend) P[b].pend->fut.wait();
    for (int b = 0; b < B; ++b) {
        if (!P[b].live) continue;                                                  // never exited: compute_matches(pred_score, 0.2) (eval/matching.py:119)
        imp_loop_pair& o = pairs[b];
        const int nb = n0v[b];
        for (int i = 0; i < nb; ++i) {
            const float ms = P[b].scored ? P[b].last_ms[i] : 0.f;
            o.mscores0[i] = ms;
            o.indices0[i] = (P[b].scored && ms > 0.2f) ? P[b].last_idx[i] : -1;
        }
        o.found = 0; o.n_iterations = n_iterations;
    }
    return IMP_OK;
}

// The EIMP loop (eval/matching.py:126-276) on B pairs in lock step, host logic included: the twin of
// imp_release_amd.matching._lockstep_group_uncertainty.  Unlike the IMP loop the pose estimates of a scored iteration cannot be deferred -
// with_uncertainty makes a pair's pool threshold 0.2 x the inlier ratio of ITS estimate (:243-247) - so they run side by side on the
// context's pose workers and the group waits for the slowest; then every live pair is pooled on its slice (pool_pair_impl), all id
// lists come back in one copy, and the kept rows are gathered into the second descriptor buffers, padded to the new largest pair.
int imp_loop_lockstep_uncertainty(imp_ctx* c, int B, const int32_t* n0v, const int32_t* n1v, int n0, int n1, const float* nkpts0,
                                  const float* scores0, const float* desc0, const float* nkpts1, const float* scores1, const float* desc1,
                                  float bin_score, int sinkhorn_iterations, int n_iterations, unsigned valid_mask, float match_ratio, int min_kpts,
                                  double error_th, double stop_pose_deg, int with_uncertainty, int n_min_tokens, int pose_threads,
                                  int pose_iterations, unsigned pose_seed, int pose_flags, imp_loop_pair_u* pairs, void* stream) {
    if (!c || !n0v || !n1v || !pairs || B < 1 || B > IMP_RAGGED_MAX) return fail(IMP_E_ARG, "imp_loop_lockstep_uncertainty: 1 .. 16 pairs, counts and pair records");
    if (!nkpts0 || !nkpts1 || !scores0 || !scores1 || !desc0 || !desc1) return fail(IMP_E_ARG, "imp_loop_lockstep_uncertainty: null input");
    if (match_ratio > 0.2f) return fail(IMP_E_ARG, "imp_loop_lockstep_uncertainty: match_ratio must be <= 0.2 (the final p = 0.2 matches are derived from the scored ones)");
    if (n_iterations < 1 || 2 * n_iterations > c->cfg.n_gnn_layers) return fail(IMP_E_ARG, "imp_loop_lockstep_uncertainty: more iterations than the model has layer pairs");
    for (int b = 0; b < B; ++b)
        if (!pairs[b].pts0 || !pairs[b].pts1 || !pairs[b].K0 || !pairs[b].K1 || !pairs[b].indices0 || !pairs[b].mscores0 || !pairs[b].kept0 || !pairs[b].kept1)
            return fail(IMP_E_ARG, "imp_loop_lockstep_uncertainty: a pair record with a null array");
    int rc = imp_set_counts(c, B, n0v, n1v);
    if (rc) return rc;
    struct Restore { imp_ctx* c; ~Restore() { c->rc.on = 0; c->rc_batch = 0; } } restore{c};
    if ((rc = check_ready(c, B, n0, n1))) return rc;
    hipStream_t st = S(stream);
    const int D = c->D;
    {   // buffers, sized for the first (largest) shape of the group
        const size_t need = (size_t)B * n0;
        if (need > c->lp_cap) {
            HIP_TRY(hipStreamSynchronize(st));
            if (c->lp_idx) (void)hipFree(c->lp_idx);
            if (c->lp_ms) (void)hipFree(c->lp_ms);
            if (c->lp_pin) (void)hipHostFree(c->lp_pin);
            c->lp_idx = nullptr; c->lp_ms = nullptr; c->lp_pin = nullptr; c->lp_cap = 0;
            const size_t cap = need < 16384 ? 16384 : need;
            HIP_TRY(hipMalloc(&c->lp_idx, cap * sizeof(int64_t)));
            HIP_TRY(hipMalloc(&c->lp_ms, cap * sizeof(float)));
            HIP_TRY(hipHostMalloc(&c->lp_pin, cap * 12));
            c->lp_cap = cap;
        }
        const size_t sneed = (size_t)B * (n0 + 1) * (n1 + 1);
        if (sneed > c->lu_scores_cap) {
            HIP_TRY(hipStreamSynchronize(st));
            if (c->lu_scores) (void)hipFree(c->lu_scores);
            c->lu_scores = nullptr; c->lu_scores_cap = 0;
            HIP_TRY(hipMalloc(&c->lu_scores, sneed * sizeof(float)));
            c->lu_scores_cap = sneed;
        }
        const size_t pneed = (size_t)B * (n0 + n1 + 2);
        if (pneed > c->lu_pool_cap) {
            HIP_TRY(hipStreamSynchronize(st));
            if (c->lu_pool) (void)hipFree(c->lu_pool);
            if (c->lu_pool_pin) (void)hipHostFree(c->lu_pool_pin);
            c->lu_pool = nullptr; c->lu_pool_pin = nullptr; c->lu_pool_cap = 0;
            HIP_TRY(hipMalloc(&c->lu_pool, pneed * sizeof(int64_t)));
            HIP_TRY(hipHostMalloc(&c->lu_pool_pin, pneed * sizeof(int64_t)));
            c->lu_pool_cap = pneed;
        }
        const int nn[2] = {n0, n1};
        for (int s = 0; s < 2; ++s) {
            const size_t aneed = (size_t)B * nn[s] * D;
            if (aneed > c->lu_alt_cap[s]) {
                HIP_TRY(hipStreamSynchronize(st));
                if (c->lu_alt[s]) (void)hipFree(c->lu_alt[s]);
                c->lu_alt[s] = nullptr; c->lu_alt_cap[s] = 0;
                HIP_TRY(hipMalloc(&c->lu_alt[s], aneed * sizeof(float)));
                c->lu_alt_cap[s] = aneed;
            }
        }
    }
    if (!c->lp_ev) HIP_TRY(hipEventCreateWithFlags(&c->lp_ev, hipEventDisableTiming));
    const bool with_pose = pose_threads > 0;
    if (with_pose && !c->lp_workers) c->lp_workers = new PoseWorkers(pose_threads < B ? B : pose_threads, c->device);
    int64_t* const h_idx = reinterpret_cast<int64_t*>(c->lp_pin);
    float* const h_ms = reinterpret_cast<float*>(c->lp_pin + c->lp_cap * sizeof(int64_t));

    struct PairU {
        bool live = true, has_last = false, scored = false, sel = false;
        double lastR[9], lastT[3];
        std::vector<float> pts[2];                      // the surviving pixel keypoints (eval/matching.py:166-174 slices them every iteration)
        std::vector<int32_t> kept[2];                   // their indices in the pair's original numbering
        int sel_n[2] = {-1, -1};                        // pool result waiting for the next iteration: kept counts (-1: side unchanged)
        std::vector<int64_t> last_idx;
        std::vector<float> last_ms;
        std::shared_ptr<LoopPose> job;
        std::vector<int> pm0, pm1;
    };
    std::vector<PairU> P(B);
    struct WaitAll {                                    // no return path may leave a pose worker behind
        std::vector<PairU>& P;
        ~WaitAll() { for (auto& q : P) if (q.job) q.job->fut.wait(); }
    } wait_all{P};
    int pad[2] = {n0, n1};
    std::vector<int> cnt[2] = {std::vector<int>(n0v, n0v + B), std::vector<int>(n1v, n1v + B)};
    for (int b = 0; b < B; ++b) {
        pairs[b].found = 0; pairs[b].n_iterations = n_iterations; pairs[b].n_kept0 = n0v[b]; pairs[b].n_kept1 = n1v[b];
        P[b].pts[0].assign(pairs[b].pts0, pairs[b].pts0 + 2 * (size_t)n0v[b]);
        P[b].pts[1].assign(pairs[b].pts1, pairs[b].pts1 + 2 * (size_t)n1v[b]);
        for (int s = 0; s < 2; ++s) { P[b].kept[s].resize(cnt[s][b]); for (int i = 0; i < cnt[s][b]; ++i) P[b].kept[s][i] = i; }
    }
    auto finish = [&](int b, const int64_t* idx, const float* ms, const unsigned char* inl, const double* R, const double* t, int n_iter) {
        imp_loop_pair_u& o = pairs[b];
        PairU& q = P[b];
        const int nb = (int)q.kept[0].size();
        if (inl) {                                      // pose exit: the inlier-filtered matches (eval/matching.py:259-269)
            for (int i = 0; i < nb; ++i) { o.indices0[i] = -1; o.mscores0[i] = ms[i]; }
            for (size_t m = 0; m < q.pm0.size(); ++m)
                if (inl[m]) o.indices0[q.pm0[m]] = q.pm1[m];
            memcpy(o.R, R, sizeof o.R); memcpy(o.t, t, sizeof o.t);
            o.found = 1;
        } else {                                        // never exited: compute_matches(pred_score, 0.2) (eval/matching.py:271)
            // (sized by the LAST SCORED iteration: if the loop ends on an unscored one after a pool, the reference returns the sliced
            // keypoints beside the older, longer match vector - so does this)
            const int ns = idx ? (int)q.last_idx.size() : nb;
            for (int i = 0; i < ns; ++i) {
                const float v = idx ? ms[i] : 0.f;
                o.mscores0[i] = v;
                o.indices0[i] = (idx && v > 0.2f) ? idx[i] : -1;
            }
            o.found = 0;
        }
        o.n_iterations = n_iter;
        o.n_indices = (!inl && idx) ? (int)q.last_idx.size() : nb;
        o.n_kept0 = nb; o.n_kept1 = (int)q.kept[1].size();
        memcpy(o.kept0, q.kept[0].data(), q.kept[0].size() * sizeof(int32_t));
        memcpy(o.kept1, q.kept[1].data(), q.kept[1].size() * sizeof(int32_t));
    };

    const float* kp[2] = {nkpts0, nkpts1};
    const float* sc[2] = {scores0, scores1};
    const float* de[2] = {desc0, desc1};
    float* cur[2] = {c->descw[0], c->descw[1]};
    float* alt[2] = {c->lu_alt[0], c->lu_alt[1]};
    if ((rc = run_kenc(c, B, pad, kp, sc, 0.f, 0.f, de, cur, st))) return rc;     // desc + enc (eval/matching.py:158-160)
    const uint8_t* nomask[2] = {nullptr, nullptr};
    bool proj_done = false;
    const size_t pool_w = (size_t)n0 + n1 + 2;                                     // int64 words of a pair's pool record: ids0 | ids1 | 4 int32 counts
    for (int it = 0; it < n_iterations; ++it) {
        bool any_sel = false;
        for (int b = 0; b < B; ++b) any_sel = any_sel || (P[b].live && P[b].sel);
        if (any_sel) {                                                             // eval/matching.py:166-174, every pair on its own rows
            int npad[2] = {1, 1};
            std::vector<int> ncnt[2] = {cnt[0], cnt[1]};
            for (int b = 0; b < B; ++b)
                for (int s = 0; s < 2; ++s) {
                    if (!P[b].live) { ncnt[s][b] = 0; continue; }
                    if (P[b].sel && P[b].sel_n[s] >= 0) ncnt[s][b] = P[b].sel_n[s];
                    if (ncnt[s][b] > npad[s]) npad[s] = ncnt[s][b];
                }
            for (int b = 0; b < B; ++b) {
                if (!P[b].live) continue;
                for (int s = 0; s < 2; ++s) {
                    const float* src = cur[s] + (size_t)b * pad[s] * D;
                    float* dst = alt[s] + (size_t)b * npad[s] * D;
                    if (P[b].sel && P[b].sel_n[s] >= 0) {
                        const int64_t* ids = c->lu_pool + (size_t)b * pool_w + (s ? n0 : 0);
                        HIP_TRY(launch_gather_rows(src, ids, dst, 1, pad[s], ncnt[s][b], D, st));
                    } else {
                        HIP_TRY(hipMemcpyAsync(dst, src, (size_t)cnt[s][b] * D * sizeof(float), hipMemcpyDeviceToDevice, st));
                    }
                }
                P[b].sel = false; P[b].sel_n[0] = P[b].sel_n[1] = -1;
            }
            for (int s = 0; s < 2; ++s) { std::swap(cur[s], alt[s]); pad[s] = npad[s]; cnt[s] = ncnt[s]; }
            for (int b = 0; b < B; ++b) { c->rc.n[0][b] = cnt[0][b]; c->rc.n[1][b] = cnt[1][b]; }
            proj_done = false;                                                     // (a projection chained before the gather saw the old rows)
        }
        const bool scored = ((valid_mask >> it) & 1u) != 0;
        for (int li = 2 * it; li <= 2 * it + 1; ++li) {
            // the last layer before a pool must not chain the next layer's projection into the q | k | v slots the pool still reads
            const bool pool_follows = scored && li == 2 * it + 1;
            const int chain = (li + 1 < c->cfg.n_gnn_layers && !pool_follows) ? li + 1 : -1;
            bool chained = false;
            const float* dr[2] = {cur[0], cur[1]};
            if ((rc = run_layer(c, li, B, pad, dr, cur, nomask, st, proj_done, chain, &chained))) return rc;
            proj_done = chained;
        }
        if (!scored) continue;
        const float* dr[2] = {cur[0], cur[1]};
        if ((rc = run_distance(c, it, B, pad, dr, c->dist, st))) return rc;
        OtBuffers o;
        bool max_done = false;
        if ((rc = run_score(c, B, pad[0], pad[1], c->dist, bin_score, sinkhorn_iterations, 1, c->lu_scores, &o, st, &max_done))) return rc;
        if (!max_done) HIP_TRY(launch_ot_maxima(B, pad[0], pad[1], 0, o, c->max0, c->arg0, c->max1, c->arg1, st));      // (a one-pair group on the streaming kernels)
        HIP_TRY(launch_mutual_matches(B, pad[0], pad[1], c->max0, c->arg0, c->max1, c->arg1, match_ratio, c->lp_idx, nullptr, c->lp_ms, nullptr,
                                      c->range_hostdev, st, &c->rc));
        const size_t nidx = (size_t)B * pad[0];
        HIP_TRY(hipMemcpyAsync(h_idx, c->lp_idx, nidx * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_ms, c->lp_ms, nidx * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c->lp_ev, st));
        HIP_TRY(hipEventSynchronize(c->lp_ev));
        if ((rc = resident_health(c))) return rc;                                  // a voided launch: the caller runs the group again
        std::vector<int> work;
        for (int b = 0; b < B; ++b) {
            PairU& q = P[b];
            if (!q.live) continue;
            const int nb = cnt[0][b];
            const int64_t* ib = h_idx + (size_t)b * pad[0];
            const float* mb = h_ms + (size_t)b * pad[0];
            q.last_idx.assign(ib, ib + nb);
            q.last_ms.assign(mb, mb + nb);
            q.scored = true;
            q.pm0.clear(); q.pm1.clear();
            for (int i = 0; i < nb; ++i)
                if (ib[i] > -1) { q.pm0.push_back(i); q.pm1.push_back((int)ib[i]); }
            if ((int)q.pm0.size() < min_kpts) { q.has_last = false; continue; }   // eval/matching.py:191-194
            if (q.pm0.empty()) continue;
            q.job.reset();
            if (with_pose) {
                auto job = std::make_shared<LoopPose>();
                const int m = (int)q.pm0.size();
                job->n = m;
                job->k0.resize(2 * (size_t)m); job->k1.resize(2 * (size_t)m); job->mask.assign(m, 0);
                for (int j = 0; j < m; ++j) {
                    job->k0[2 * j] = q.pts[0][2 * (size_t)q.pm0[j]]; job->k0[2 * j + 1] = q.pts[0][2 * (size_t)q.pm0[j] + 1];
                    job->k1[2 * j] = q.pts[1][2 * (size_t)q.pm1[j]]; job->k1[2 * j + 1] = q.pts[1][2 * (size_t)q.pm1[j] + 1];
                }
                job->fut = job->done.get_future();
                const double* K0 = pairs[b].K0; const double* K1 = pairs[b].K1;
                const int dev = c->device;
                c->lp_workers->submit([job, K0, K1, error_th, pose_iterations, pose_seed, pose_flags, dev](hipStream_t ps) {
                    job->rc = imp_estimate_pose(job->k0.data(), job->k1.data(), job->n, K0, K1, error_th, pose_iterations, pose_seed, dev, job->E, job->R,
                                                job->t, job->mask.data(), nullptr, &job->ninl, pose_flags, ps);
                    job->done.set_value();
                });
                q.job = job;
            }
            work.push_back(b);
        }
        std::vector<std::pair<int, float>> to_pool;
        for (int b : work) {
            PairU& q = P[b];
            std::shared_ptr<LoopPose> job = q.job;
            q.job.reset();
            bool have = false;
            if (job) { job->fut.wait(); have = job->rc == 0; }
            double inlier_ratio = 0.0;
            if (have) {
                int s = 0;
                for (unsigned char v : job->mask) s += v ? 1 : 0;
                inlier_ratio = (double)s / (double)q.pm0.size();
            }
            double diff_R = INFINITY, diff_t = INFINITY;
            if (it >= 1 && have && q.has_last) { diff_R = loop_angle_mat(q.lastR, job->R); diff_t = loop_angle_vec(q.lastT, job->t); }
            q.has_last = have;
            if (have) { memcpy(q.lastR, job->R, sizeof q.lastR); memcpy(q.lastT, job->t, sizeof q.lastT); }
            const double pose_diff = diff_R > diff_t ? diff_R : diff_t;
            if (stop_pose_deg >= 0.0 && pose_diff <= stop_pose_deg) {              // eval/matching.py:259-269
                finish(b, q.last_idx.data(), q.last_ms.data(), job->mask.data(), job->R, job->t, it + 1);
                q.live = false;
                c->rc.n[0][b] = 0; c->rc.n[1][b] = 0;
                cnt[0][b] = cnt[1][b] = 0;
                continue;
            }
            // (the reference pools before it takes the exit test; a pair that exits never uses the result)
            const float th = (with_uncertainty && inlier_ratio != 0.0) ? (float)(0.2 * inlier_ratio) : 0.2f;      // eval/matching.py:243-247
            to_pool.emplace_back(b, th);
        }
        bool any = false;
        for (int b = 0; b < B; ++b) any = any || P[b].live;
        if (!any) break;
        if (to_pool.empty() || it + 1 >= n_iterations) continue;
        for (auto& pt : to_pool) {
            const int b = pt.first;
            const int m[2] = {cnt[0][b], cnt[1][b]};
            int64_t* rec = c->lu_pool + (size_t)b * pool_w;
            const float* sb = c->lu_scores + (size_t)b * (pad[0] + 1) * (pad[1] + 1);
            if ((rc = pool_pair_impl(c, b, m, pad, sb, pt.second, n_min_tokens, rec, rec + n0, reinterpret_cast<int32_t*>(rec + n0 + n1), st))) return rc;
        }
        HIP_TRY(hipMemcpyAsync(c->lu_pool_pin, c->lu_pool, (size_t)B * pool_w * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c->lp_ev, st));
        HIP_TRY(hipEventSynchronize(c->lp_ev));
        for (auto& pt : to_pool) {
            const int b = pt.first;
            PairU& q = P[b];
            const int64_t* rec = c->lu_pool_pin + (size_t)b * pool_w;
            const int32_t* cn = reinterpret_cast<const int32_t*>(rec + n0 + n1);
            const int got[2] = {cn[0], cn[2]};
            for (int s = 0; s < 2; ++s) {
                if (got[s] < 0) continue;                                          // side skipped, or nothing confident: unchanged
                const int64_t* ids = rec + (s ? n0 : 0);
                std::vector<float> np(2 * (size_t)got[s]);
                std::vector<int32_t> nk(got[s]);
                for (int i = 0; i < got[s]; ++i) {
                    const int64_t id = ids[i];
                    np[2 * (size_t)i] = q.pts[s][2 * (size_t)id]; np[2 * (size_t)i + 1] = q.pts[s][2 * (size_t)id + 1];
                    nk[i] = q.kept[s][(size_t)id];
                }
                q.pts[s].swap(np); q.kept[s].swap(nk);
                q.sel_n[s] = got[s];
                q.sel = true;
            }
        }
    }
    for (int b = 0; b < B; ++b)
        if (P[b].live) finish(b, P[b].scored ? P[b].last_idx.data() : nullptr, P[b].last_ms.data(), nullptr, nullptr, nullptr, n_iterations);
    return IMP_OK;
}

int imp_ctx_option(imp_ctx* c, const char* name, long value) {
    if (!c) return fail(IMP_E_ARG, "imp_ctx_option: no context");
    if (ctx_option(c, name, value) != IMP_OK) return fail(IMP_E_ARG, std::string("imp_ctx_option: unknown option '") + (name ? name : "") + "'");
    return IMP_OK;
}

int imp_debug_hold_cus(int device, int workgroups, int microseconds, void* stream) {
    if (workgroups < 1 || workgroups > 1024 || microseconds < 1 || microseconds > 1000000) return fail(IMP_E_ARG, "imp_debug_hold_cus: 1..1024 workgroups, 1..1e6 microseconds");
    HIP_TRY(hipSetDevice(device));
    const size_t lds = 96 * 1024;
    if (hipError_t e = imp_grant_dynamic_lds((const void*)hold_cus_kernel, lds)) return fail(IMP_E_HIP, hipGetErrorString(e));
    hipLaunchKernelGGL(hold_cus_kernel, dim3(workgroups), dim3(256), lds, S(stream), (unsigned long long)microseconds * 100ull, (unsigned*)nullptr);
    HIP_TRY(hipGetLastError());
    return IMP_OK;
}

int imp_set_counts(imp_ctx* c, int batch, const int32_t* n0, const int32_t* n1) {
    if (!c) return fail(IMP_E_ARG, "null context");
    if (!n0 && !n1) { c->rc.on = 0; c->rc_batch = 0; return IMP_OK; }
    if (!n0 || !n1 || batch < 1 || batch > IMP_RAGGED_MAX)
        return fail(IMP_E_ARG, "imp_set_counts: 1 .. " + std::to_string(IMP_RAGGED_MAX) + " pairs, both count arrays (or both null to clear)");
    for (int b = 0; b < batch; ++b) {
        if (n0[b] < 0 || n1[b] < 0 || ((n0[b] == 0) != (n1[b] == 0)))
            return fail(IMP_E_ARG, "imp_set_counts: counts must be positive, or 0 for BOTH images of a retired pair");
        c->rc.n[0][b] = n0[b]; c->rc.n[1][b] = n1[b];
    }
    for (int b = batch; b < IMP_RAGGED_MAX; ++b) c->rc.n[0][b] = c->rc.n[1][b] = 0;
    c->rc.on = 1;
    c->rc_batch = batch;
    return IMP_OK;
}

int imp_op_linear(imp_ctx* c, int M, int N, int K, const float* x, const float* W, const float* bias, float* y,
                  void* stream) {
    if (!c || !x || !W || !y || K % 32) return fail(IMP_E_ARG, "imp_op_linear: bad argument (K % 32 == 0)");
    HIP_TRY(hipSetDevice(c->device));
    GemmParams p = gemm_defaults(c, K);
    p.nside = 1;
    GemmSide& g = p.side[0];
    g.A = x; g.W = W; g.C = y; g.M = M; g.N = N;
    p.bias = bias; p.lda = K; p.ldw = K; p.ldc = N;
    HIP_TRY(launch_gemm_f32(p, 1, S(stream)));
    return IMP_OK;
}

int imp_op_layer_gemm(imp_ctx* c, int B, int M, int N, int K, int ksplit, const float* x, const float* x2, const float* W, const float* bias,
                      const float* residual, const float* stats_in, float* y, float* stats_out, const float* W2, const float* bias2, int N2,
                      float* y2, int pass_split, void* stream) {
    if (!c || !x || !W || !y || B < 1 || M < 1 || !gemm_wf_supported(K, N) || ksplit < 0 || ksplit > K || (ksplit < K && !x2) || ksplit % 4)
        return fail(IMP_E_ARG, "imp_op_layer_gemm: bad argument (K 256 / 512, N % 128 == 0)");
    if (W2 && (!y2 || !bias2 || N != 256 || K != 512 || !stats_in || stats_out || !gemm_wf_supported(256, N2) || N2 < 256))
        return fail(IMP_E_ARG, "imp_op_layer_gemm: the chained projection needs K = 512, N = 256, stats_in, N2 % 128 == 0, N2 >= 256");
    if (stats_out && (stats_in || N > 512)) return fail(IMP_E_ARG, "imp_op_layer_gemm: stats_out excludes stats_in; N <= 512");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = S(stream);
    HIP_TRY(hipStreamSynchronize(st));
    auto pack = [&](const float* Wd, int n, int k, _Float16** out) -> int {
        std::vector<float> h((size_t)n * k);
        HIP_TRY(hipMemcpy(h.data(), Wd, h.size() * 4, hipMemcpyDeviceToHost));
        std::vector<_Float16> f((size_t)n * k * 2);
        wf_pack(h.data(), n, k, f.data());
        HIP_TRY(hipMalloc(out, f.size() * 2));
        HIP_TRY(hipMemcpy(*out, f.data(), f.size() * 2, hipMemcpyHostToDevice));
        return IMP_OK;
    };
    _Float16 *wf = nullptr, *wf2 = nullptr;
    float* part = nullptr;
    unsigned* cnt = nullptr;
    int rc = pack(W, N, K, &wf);
    if (!rc && W2) rc = pack(W2, N2, 256, &wf2);
    if (rc) return rc;
    const int tiles = (M + 63) / 64;
    if (stats_out) {
        HIP_TRY(hipMalloc(&part, (size_t)B * tiles * N * 2 * sizeof(float)));
        HIP_TRY(hipMalloc(&cnt, (size_t)B * WF_MAX_PSPLIT * sizeof(unsigned)));
        HIP_TRY(hipMemset(cnt, 0, (size_t)B * WF_MAX_PSPLIT * sizeof(unsigned)));
        HIP_TRY(hipDeviceSynchronize());
    }
    WfParams p = wf_defaults();
    p.K = K; p.ksplit = ksplit; p.N = N; p.nside = 1;
    WfSide& g = p.side[0];
    g.A = x; g.A2 = ksplit < K ? x2 : nullptr; g.C = y; g.R = residual; g.M = M;
    g.sA_b = (long)M * ksplit; g.sA2_b = (long)M * (K - ksplit); g.sC_b = (long)M * N; g.sR_b = (long)M * N;
    g.in_stats = stats_in; g.out_stats = part; g.fin_stats = stats_out;
    g.C2 = y2; g.sC2_b = (long)M * N2;
    p.Wf_ = wf; p.bias = bias; p.lda = ksplit; p.lda2 = K - ksplit; p.ldc = N; p.ldr = N;
    p.norm_eps = 1e-3f; p.stat_cnt = cnt;
    p.Wf2_ = wf2; p.bias2 = bias2; p.N2 = N2; p.ldc2 = N2;
    p.pass_split = pass_split > 1 ? pass_split : 1;
    const hipError_t e = launch_gemm_wf(p, B, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(wf); (void)hipFree(wf2); (void)hipFree(part); (void)hipFree(cnt);
    HIP_TRY(e);
    HIP_TRY(e2);
    return IMP_OK;
}

int imp_op_fused_mlp(imp_ctx* c, int B, int M, const float* x, const float* a, const float* W0, const float* b0, const float* W3, const float* b3,
                     const float* W2, const float* b2, int N2, float* y, float* y2, int fake, void* stream) {
    if (!c || !x || !a || !W0 || !b0 || !W3 || !b3 || !y || B < 1 || M < 1) return fail(IMP_E_ARG, "imp_op_fused_mlp: bad argument");
    if (W2 && (!y2 || !b2 || !gemm_wf_supported(256, N2) || N2 < 256)) return fail(IMP_E_ARG, "imp_op_fused_mlp: the chained projection needs N2 % 128 == 0, N2 >= 256");
    const int tiles = (M + 63) / 64;
    if ((long)B * tiles > c->num_cus) return fail(IMP_E_ARG, "imp_op_fused_mlp: more 64-row tiles than CUs (the tiles of a launch wait for each other)");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = S(stream);
    HIP_TRY(hipStreamSynchronize(st));
    auto pack = [&](const float* Wd, int n, int k, _Float16** out) -> int {
        std::vector<float> h((size_t)n * k);
        HIP_TRY(hipMemcpy(h.data(), Wd, h.size() * 4, hipMemcpyDeviceToHost));
        std::vector<_Float16> fr((size_t)n * k * 2);
        wf_pack(h.data(), n, k, fr.data());
        HIP_TRY(hipMalloc(out, fr.size() * 2));
        HIP_TRY(hipMemcpy(*out, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        return IMP_OK;
    };
    _Float16 *wf0 = nullptr, *wf3 = nullptr, *wf2 = nullptr;
    float *rec = nullptr, *fin = nullptr;
    int* status = nullptr;
    int rc = pack(W0, 512, 512, &wf0);
    if (!rc) rc = pack(W3, 256, 512, &wf3);
    if (!rc && W2) rc = pack(W2, N2, 256, &wf2);
    if (rc) return rc;
    const size_t rec_b = (size_t)B * tiles * 512 * 16, fin_b = (size_t)B * 512 * 16;
    HIP_TRY(hipMalloc(&rec, rec_b));
    HIP_TRY(hipMalloc(&fin, fin_b));
    HIP_TRY(hipMalloc(&status, 64));
    HIP_TRY(hipMemset(rec, 0, rec_b));
    HIP_TRY(hipMemset(fin, 0, fin_b));
    HIP_TRY(hipMemset(status, 0, 64));
    HIP_TRY(hipDeviceSynchronize());
    WfParams p = wf_defaults();
    p.K = 512; p.ksplit = 256; p.N = 512; p.nside = 1;
    WfSide& g = p.side[0];
    g.A = x; g.A2 = a; g.C = y; g.R = x; g.M = M;
    g.sA_b = g.sA2_b = g.sC_b = g.sR_b = (long)M * 256;
    g.C2 = y2; g.sC2_b = (long)M * N2;
    p.norm_eps = 1e-3f;
    p.Wf_ = wf0; p.bias = b0; p.lda = p.lda2 = 256; p.ldc = 256; p.ldr = 256;
    p.Wf2_ = wf2; p.bias2 = b2; p.N2 = N2; p.ldc2 = N2;
    p.pass_split = 1;
    WfFused f;
    memset(&f, 0, sizeof f);
    f.Wf3_ = wf3; f.bias3 = b3; f.rec[0] = rec; f.fin[0] = fin; f.tag = 1u; f.status = status; f.fake = fake;
    int lrc = IMP_OK;
    hipError_t e = hipSuccess;
    SpinGate* gate = spin_gate(c->device);
    if (!gate) lrc = fail(IMP_E_HIP, "spin gate: cannot create events");
    if (!lrc) lrc = spin_enter(gate, st);
    if (!lrc) {
        e = launch_gemm_wf_fused(p, f, B, st);
        lrc = spin_leave(gate, st);
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    int hst = 0;
    (void)hipMemcpy(&hst, status, 4, hipMemcpyDeviceToHost);
    (void)hipFree(wf0); (void)hipFree(wf3); (void)hipFree(wf2); (void)hipFree(rec); (void)hipFree(fin); (void)hipFree(status);
    if (lrc) return lrc;
    HIP_TRY(e);
    HIP_TRY(e2);
    if (hst) return fail(IMP_E_RESIDENT, "imp_op_fused_mlp: the statistics exchange timed out (outputs are NaN)");
    return IMP_OK;
}

int imp_op_attention(imp_ctx* c, int batch, int nq, int nk, int dim, const float* qkv_q, const float* qkv_kv,
                     const uint8_t* key_mask, float* out, float* lse, void* stream) {
    if (!c || !qkv_q || !qkv_kv || !out || (dim != 256 && dim != 128)) return fail(IMP_E_ARG, "imp_op_attention: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    AttnParams a;
    memset(&a, 0, sizeof a);
    a.nside = 1; a.ldq = a.ldk = 3 * dim; a.ldo = dim; a.dh = dim / IMP_NUM_HEADS;
    AttnSide& g = a.side[0];
    g.q = qkv_q; g.k = qkv_kv + dim; g.v = qkv_kv + 2 * dim; g.out = out; g.lse = lse; g.kmask = key_mask;
    g.sq_b = (long)nq * 3 * dim; g.sk_b = (long)nk * 3 * dim; g.so_b = (long)nq * dim;
    g.nq = nq; g.nk = nk;
    if (int arc = launch_attention(c, a, batch, S(stream))) return arc;
    return IMP_OK;
}

int imp_time_attention(imp_ctx* c, int batch, int n, int reps, float* ms, void* stream) {
    return imp_time_attention_clock(c, batch, n, reps, ms, nullptr, stream);
}

int imp_time_attention_clock(imp_ctx* c, int batch, int n, int reps, float* ms, float* sclk_mhz, void* stream) {
    int rc = check_ready(c, batch, n, n);
    if (rc) return rc;
    if (!ms || reps < 1) return fail(IMP_E_ARG, "imp_time_attention: bad argument");
    hipStream_t st = S(stream);
    const int D = c->D;
    AttnParams a;
    memset(&a, 0, sizeof a);
    a.nside = 2; a.ldq = a.ldk = 3 * D; a.ldo = D; a.dh = c->dh;
    for (int s = 0; s < 2; ++s) {
        AttnSide& g = a.side[s];
        g.q = c->qkv[0][s]; g.k = c->qkv[0][s] + D; g.v = c->qkv[0][s] + 2 * D; g.out = c->attn_out[s];
        g.sq_b = g.sk_b = (long)n * 3 * D; g.so_b = (long)n * D; g.nq = g.nk = n;
    }
    // the launch the product makes: with K / V as split-half images (default in f16x3 mode, D = 256) the q | k | v slots are first filled by
    // the layer-0 projection kernel itself from whatever descriptors the workspace holds (option kv_image = 0: fp32 k | v, split while staging)
    {
        const bool kvp = c->prec == 1 && c->kv_image && c->use_wf && D == 256 && !c->layers.empty() && c->layers[0].proj_wf && !c->layers[0].shared;
        if (c->prec == 1 && D == 256 && !c->layers.empty() && c->layers[0].proj_wf && !c->layers[0].shared) {
            const GnnLayer& L = c->layers[0];
            WfParams p = wf_defaults();
            p.K = D; p.ksplit = D; p.N = L.proj.out; p.nside = 2;
            for (int s = 0; s < 2; ++s) {
                WfSide& g = p.side[s];
                g.A = c->descw[s]; g.M = n; g.C = c->qkv[0][s];
                g.sA_b = (long)n * D; g.sC_b = (long)n * 3 * D;
            }
            p.Wf_ = L.proj_wf; p.bias = L.proj.b; p.lda = D; p.ldc = 3 * D;
            if (kvp) p.kv_image_col = D;
            p.pass_split = wf_pass_split(c, (long)batch * 2 * ((n + 63) / 64), p.N);
            HIP_TRY(launch_gemm_wf(p, batch, st));
        }
        a.kv_planes = kvp ? 1 : 0;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    if (int arc = launch_attention(c, a, batch, st)) return arc;   // warm
    unsigned long long* probe = nullptr;
    if (sclk_mhz) {                                                // workgroup 0 of every timed launch adds its lifetime in shader cycles / 100 MHz ticks
        HIP_TRY(hipMalloc(&probe, 2 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(probe, 0, 2 * sizeof(unsigned long long), st));
        a.clk_probe = probe;
    }
    HIP_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        if (int arc = launch_attention(c, a, batch, st)) return arc;
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / reps;
    if (sclk_mhz) {
        unsigned long long h[2] = {0, 0};
        HIP_TRY(hipMemcpy(h, probe, sizeof h, hipMemcpyDeviceToHost));
        (void)hipFree(probe);
        *sclk_mhz = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.f;
    }
    return IMP_OK;
}

int imp_time_sinkhorn(imp_ctx* c, int batch, int n, int iterations, float* ms, void* stream) {
    int rc = check_ready(c, batch, n, n);
    if (rc) return rc;
    if (!ms || iterations < 1) return fail(IMP_E_ARG, "imp_time_sinkhorn: bad argument");
    hipStream_t st = S(stream);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float t = 0.f;
    int nch, rpw, G, local = 0;
    rc = c->ot_resident ? ensure_resident_buffers(c, batch) : 1;
    if (rc < 0) return rc;
    if (rc == 0 && plan_resident(c, batch, n, n, c->num_cus, &nch, &rpw, &G, &local)) {
        // resident path: per-iteration time = (launch with T iterations - launch with 0 iterations) / T, both on `st`
        // (nothing else runs during a timing call), events on the stream the kernel is launched on
        OtResidentParams p;
        memset(&p, 0, sizeof p);
        p.dist = c->dist; p.B = batch; p.n0 = n; p.n1 = n; p.G = G; p.bin = 1.f;
        p.xpart = c->xpart; p.xv = c->xv; p.xmax = c->xmax; p.status = c->xstatus;
        p.local = local; p.xhalf = c->xhalf;
        float tt[2];
        for (int k = 0; k < 2; ++k) {
            p.T = k ? iterations : 0;
            p.tag_base = resident_tags(c, p.T);
            resident_health_params(c, &p);
            HIP_TRY(launch_ot_resident(p, nch, rpw, st));   // warm
            HIP_TRY(hipEventRecord(e0, st));
            for (int r = 0; r < 3; ++r) {
                p.tag_base = resident_tags(c, p.T);
                resident_health_params(c, &p);
                HIP_TRY(launch_ot_resident(p, nch, rpw, st));
            }
            HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(&tt[k], e0, e1));
        }
        t = (tt[1] - tt[0]) / 3.f / iterations;
        if (c->probe_prof) {              // probe: phase cycles of workgroup 0 over one launch of `iterations`
            unsigned long long* dprof = nullptr;
            HIP_TRY(hipMalloc(&dprof, 6 * sizeof(unsigned long long)));
            p.prof = dprof; p.T = iterations; p.tag_base = resident_tags(c, p.T);
            resident_health_params(c, &p);
            HIP_TRY(launch_ot_resident(p, nch, rpw, st));
            unsigned long long hp[6];
            HIP_TRY(hipMemcpy(hp, dprof, sizeof hp, hipMemcpyDeviceToHost));
            (void)hipFree(dprof);
            fprintf(stderr, "[otr prof] B=%d n=%d G=%d iterations=%d cycles per iteration (100 MHz-domain counter x?): A %.0f  B %.0f  wait+stage %.0f  reduce+store %.0f  wait+read v %.0f  vsum %.0f\n",
                    batch, n, G, iterations, (double)hp[0] / iterations, (double)hp[1] / iterations, (double)hp[2] / iterations,
                    (double)hp[3] / iterations, (double)hp[4] / iterations, (double)hp[5] / iterations);
        }
    } else {
        OtBuffers o;
        ot_layout(c, n, n, &o);
        HIP_TRY(launch_ot_iterations(batch, n, n, 2, o, st));   // warm
        HIP_TRY(hipEventRecord(e0, st));
        HIP_TRY(launch_ot_iterations(batch, n, n, iterations, o, st));
        HIP_TRY(hipEventRecord(e1, st));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        t /= iterations;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t;          // milliseconds per Sinkhorn ITERATION
    return IMP_OK;
}

// probe: average time of ONE of the three layer GEMMs of layer 0 on the context's workspace (which: 0 QKV, 1 MLP0, 2 MLP3, 3 MLP3 chained with QKV),
// dbg == -1: the gemm_f32.hip kernel; dbg <= -2: gemm_wf.hip with its probe switches (-dbg - 2)
int imp_time_layer_gemm(imp_ctx* c, int batch, int n, int which, int dbg, int reps, float* ms, void* stream) {
    int rc = check_ready(c, batch, n, n);
    if (rc) return rc;
    if (!ms || reps < 1 || which < 0 || which > 4 || c->layers.empty()) return fail(IMP_E_ARG, "imp_time_layer_gemm: bad argument");
    if (which == 4 && (dbg != -2 || c->D != 256 || !c->layers[0].mlp0f_wf || !c->layers[0].mlp3_wf || !c->layers[0].proj_wf || !c->fx_rec[0] ||
                       (long)batch * 2 * ((n + 63) / 64) > c->num_cus))
        return fail(IMP_E_ARG, "imp_time_layer_gemm: which = 4 (the fused layer MLP + projection) needs dbg = -2, D = 256 and at most one 64-row tile per CU");
    if (which == 3 && dbg > -2) return fail(IMP_E_ARG, "imp_time_layer_gemm: which = 3 (MLP3 chained with the projection) exists only in gemm_wf.hip (dbg <= -2)");
    hipStream_t st = S(stream);
    const int D = c->D;
    const GnnLayer& L = c->layers[0];
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto launch = [&]() -> hipError_t {
        if (which == 4) {                                       // steps 4 + 5 + next projection as the one fused launch of run_layer
            const int nn[2] = {n, n};
            const float* de[2] = {c->descw[0], c->descw[1]};
            float* ou[2] = {c->mdesc[0], c->mdesc[1]};
            return launch_fused_layer(c, L, batch, nn, de, ou, &L, c->qkv[0], c->kv_image != 0, st) == IMP_OK ? hipSuccess : hipErrorUnknown;
        }
        if (dbg <= -2) {                                        // gemm_wf.hip; -3 / -4 / -6: its probe switches 1 / 2 / 4
            WfParams p = wf_defaults();
            p.nside = 2;
            p.K = which == 0 ? D : 2 * D; p.ksplit = which == 1 ? D : p.K;
            for (int s = 0; s < 2; ++s) {
                WfSide& g = p.side[s];
                g.M = n;
                if (which == 0) { g.A = c->descw[s]; g.C = c->qkv[0][s]; g.sA_b = (long)n * D; g.sC_b = (long)n * 3 * D; }
                if (which == 1) { g.A = c->descw[s]; g.A2 = c->attn_out[s]; g.C = c->hid[s]; g.sA_b = g.sA2_b = (long)n * D; g.sC_b = (long)n * 2 * D; g.out_stats = c->stats[s];
                                  g.fin_stats = c->nstat[s]; }
                if (which >= 2) { g.A = c->hid[s]; g.C = c->mdesc[s]; g.R = c->descw[s]; g.sA_b = (long)n * 2 * D; g.sC_b = (long)n * D; g.sR_b = (long)n * D;
                                  g.in_stats = c->nstat[s]; }
                if (which == 3) { g.C2 = c->qkv[0][s]; g.sC2_b = (long)n * 3 * D; }
            }
            p.norm_eps = 1e-3f; p.stat_cnt = c->stat_cnt;
            if (which == 0) { p.Wf_ = L.proj_wf; p.N = 3 * D; p.bias = L.proj.b; p.lda = D; p.ldc = 3 * D; }
            if (which == 1) { p.Wf_ = L.mlp0f_wf; p.N = 2 * D; p.bias = L.mlp0f.b; p.lda = p.lda2 = D; p.ldc = 2 * D; }
            if (which >= 2) { p.Wf_ = L.mlp3_wf; p.N = D; p.bias = L.mlp3.b; p.lda = 2 * D; p.ldc = D; p.ldr = D; }
            if (which == 3) { p.Wf2_ = L.proj_wf; p.bias2 = L.proj.b; p.N2 = L.proj.out; p.ldc2 = 3 * D; }
            if (!p.Wf_ || (which == 3 && !p.Wf2_)) return hipErrorInvalidValue;
            p.pass_split = which == 3 ? 1 : wf_pass_split(c, (long)batch * 2 * ((n + 63) / 64), p.N);
            p.dbg = -dbg - 2;
            return launch_gemm_wf(p, batch, st);
        }
        if (dbg < 0) {
            GemmParams p = gemm_defaults(c, which == 0 ? D : 2 * D);
            if (which == 1) p.ksplit = D;
            for (int s = 0; s < 2; ++s) {
                GemmSide& g = p.side[s];
                g.M = n;
                if (which == 0) { g.A = c->descw[s]; g.W = L.proj.W; g.C = c->qkv[0][s]; g.N = 3 * D; g.sA_b = (long)n * D; g.sC_b = (long)n * 3 * D; }
                if (which == 1) { g.A = c->descw[s]; g.A2 = c->attn_out[s]; g.W = L.mlp0f.W; g.C = c->hid[s]; g.N = 2 * D; g.sA_b = (long)n * D; g.sC_b = (long)n * 2 * D; g.out_stats = c->stats[s]; }
                if (which == 2) { g.A = c->hid[s]; g.W = L.mlp3.W; g.C = c->mdesc[s]; g.R = c->descw[s]; g.N = D; g.sA_b = (long)n * 2 * D; g.sC_b = (long)n * D; g.sR_b = (long)n * D; g.in_stats = c->nstat[s]; }
            }
            if (which == 0) { p.bias = L.proj.b; p.lda = D; p.ldw = D; p.ldc = 3 * D; }
            if (which == 1) { p.flags |= GEMM_EPI_STATS; p.bias = L.mlp0f.b; p.lda = D; p.lda2 = D; p.ldw = 2 * D; p.ldc = 2 * D; }
            if (which == 2) { p.flags = GEMM_PRO_NORM; p.bias = L.mlp3.b; p.lda = 2 * D; p.ldw = 2 * D; p.ldc = D; p.ldr = D; }
            return launch_gemm_f32(p, batch, st);
        }
        return hipErrorInvalidValue;                            // (dbg >= 0 selected the retired planes kernel: tools/probe/gemm_planes.hip)
    };
    HIP_TRY(launch());
    if (which == 4 && c->probe_prof) {       // probe (a -DWF_PROFILE build of the library): phase cycle stamps of every workgroup of one fused launch
        const int grid = batch * 2 * ((n + 63) / 64);
        unsigned long long* dprof = nullptr;
        HIP_TRY(hipMalloc(&dprof, (size_t)grid * 12 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(dprof, 0, (size_t)grid * 12 * sizeof(unsigned long long)));
        const int nn[2] = {n, n};
        const float* de[2] = {c->descw[0], c->descw[1]};
        float* ou[2] = {c->mdesc[0], c->mdesc[1]};
        for (int rep = 0; rep < 3; ++rep) {
            if (int prc = launch_fused_layer(c, L, batch, nn, de, ou, &L, c->qkv[0], c->kv_image != 0, st, dprof)) return prc;
            HIP_TRY(hipStreamSynchronize(st));
            std::vector<unsigned long long> h((size_t)grid * 12);
            HIP_TRY(hipMemcpy(h.data(), dprof, h.size() * 8, hipMemcpyDeviceToHost));
            for (int g = 0; g < 2; ++g) {
                double a[6] = {}, mx[6] = {};
                for (int i = 0; i < grid; ++i)
                    for (int j = 0; j < 6; ++j) { const double v = (double)h[((size_t)i * 2 + g) * 6 + j]; a[j] += v / grid; if (v > mx[j]) mx[j] = v; }
                fprintf(stderr, "[gemm_wf_fused B=%d n=%d, %d workgroups, group %d] mean (max) cycles: staging %.0f (%.0f)  mlp.0 K loops + block stats %.0f (%.0f)  exchange %.0f (%.0f)  "
                                "normalise -> planes %.0f (%.0f)  mlp.3 + epilogue %.0f (%.0f)  chained projection %.0f (%.0f)  total %.0f\n",
                        batch, n, grid, g, a[0], mx[0], a[1], mx[1], a[2], mx[2], a[3], mx[3], a[4], mx[4], a[5], mx[5], a[0] + a[1] + a[2] + a[3] + a[4] + a[5]);
            }
        }
        (void)hipFree(dprof);
    }
    HIP_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) HIP_TRY(launch());
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / reps;
    return IMP_OK;
}

int imp_resident_status(imp_ctx* c, int* status, int* used) {
    if (!c || !status) return fail(IMP_E_ARG, "imp_resident_status: null argument");
    HIP_TRY(hipSetDevice(c->device));
    *status = 0;
    if (used) *used = c->xstatus != nullptr;
    if (c->xstatus) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(status, c->xstatus, sizeof(int), hipMemcpyDeviceToHost));
        if (!*status) *status = *static_cast<volatile int*>(c->xstatus_host);
    }
    return IMP_OK;
}

int imp_resident_health(imp_ctx* c, int* timeouts, int* level) {
    if (!c) return fail(IMP_E_ARG, "imp_resident_health: null context");
    const int rc = resident_health(c);
    if (timeouts) *timeouts = c->resident_timeouts;
    if (level) *level = c->ot_degrade;
    return rc;
}

int imp_resident_repaired(imp_ctx* c) { return c ? c->resident_repaired : -1; }

int imp_resident_postmortem(imp_ctx* c, int32_t* out, int n) {
    if (!c || !out || n < 1) return fail(IMP_E_ARG, "imp_resident_postmortem: bad argument");
    if (!c->postmortem_valid) return 0;
    for (int i = 0; i < n && i < 40; ++i) out[i] = c->postmortem[i];
    return 1;
}

int imp_range_events(imp_ctx* c) { return c ? c->range_events : -1; }
int imp_set_range_recovery(imp_ctx* c, int on) {
    if (!c) return fail(IMP_E_ARG, "imp_set_range_recovery: null context");
    c->range_recover = on ? 1 : 0;
    return IMP_OK;
}
int imp_range_recovered(imp_ctx* c) { return c ? c->range_recovered : -1; }
int imp_range_take(imp_ctx* c, int recovered) {
    if (!c || !c->range_host) return 0;
    if (recovered == 2) { c->range_recovered += 1; return 0; }       // (the caller repaired a pass whose event an entry point had already reported: IMP_E_RANGE mid-pass)
    if (c->xstatus_host && *static_cast<volatile int*>(c->xstatus_host)) return 0;      // a voided resident launch is the health check's business
    if (!*static_cast<volatile int*>(c->range_host)) return 0;
    *static_cast<volatile int*>(c->range_host) = 0;
    c->range_events += 1;
    if (recovered) c->range_recovered += 1;
    return 1;
}

int imp_tag_wraps(imp_ctx* c) { return c ? c->tag_wraps : -1; }

int imp_set_resident_verify(imp_ctx* c, int on) {
    if (!c) return fail(IMP_E_ARG, "imp_set_resident_verify: null context");
    c->ot_verify = on != 0;
    return IMP_OK;
}

}  // extern "C"
