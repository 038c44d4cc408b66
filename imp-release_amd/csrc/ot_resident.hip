// Chip-resident Sinkhorn (nets/layers.py:27-46 + nets/gm.py:305-307), gfx950.
//
// The streaming path (ot.hip) reads the (N+1)x(M+1) matrix from HBM/LLC once per iteration through two dependent
// launches: 19 us per iteration at B = 4, N = 2048, 200 launches per pair batch.  But the matrix FITS ON THE CHIP: 4 pairs
// x 2048 x 2048 fp32 = 64 MiB = 128 VGPRs of every lane of a 512-thread workgroup on each of the 256 CUs.  This kernel
// keeps it there for all T iterations: ONE launch computes the row softmax of the dustbin-augmented matrix straight from
// the distance matrix into registers, runs the T iterations exchanging only vectors between workgroups, and finishes with
// the scores and the row / column maxima - P never exists in memory.
//
// Decomposition: a pair's real rows are split over G workgroups (8 waves x RPW rows each); a lane owns NCH float4 chunks
// of each of its rows (columns 4 (lane + 64 c) .. +3).  The dustbin COLUMN is one scalar per row; the dustbin ROW is the
// constant 1 / (n1 + 1) (softmax of a constant row) and is never stored.  One iteration:
//   A  u_i = 1 / (P_i . v + eps) for the own rows, column partials += P_i u_i          (registers + v from LDS)
//   B  8 waves' partials summed in LDS -> the workgroup's partial vector -> memory      (write-through stores)
//   -- group barrier --
//   C  every workgroup sums ITS SLICE of columns over the G partial vectors (fixed order), adds the dustbin row,
//      v_j = c_j / (. + eps) -> memory
//   -- group barrier --
//   D  every workgroup reads v
// Exchange protocol: there is NO barrier in the loop.  Every exchanged float travels as an 8-byte granule {value, tag}
// (an aligned 8-byte store / load is single-copy atomic), tag = a per-launch base + the exchange's sequence number; a
// consumer polls the granules it needs (sc1 = agent-scope loads that bypass the non-coherent caches) until the tag is the
// one it waits for, so a step costs one store-to-load propagation instead of store-acknowledge + atomic barrier + load
// (measured on MI355X, B = 4, N = 2048: 10.5 us per iteration with fence-free atomic barriers - themselves 1.0-1.3 us
// each against ~8 us with __threadfence, tools/probe/barrier2_probe.hip - and see DESIGN.md for this protocol).  A
// buffer is reused by the next iteration only after everyone has consumed it: a workgroup writes partial(t+1) after it
// read v(t), which exists only once every slice owner has read all of partial(t); likewise for v.  Polls are bounded:
// a time-out raises *status (device flag, polled by every waiter) AND *host_status (a word in mapped host memory the
// library reads without synchronising at its next entry, context.hip resident_health), every later wait of the launch
// falls through, and a workgroup that saw a time-out POISONS its outputs (maxima NaN, arg-maxima 0x7fffffff -> the match
// kernel reports mscore NaN / index -1; first score of each own row NaN): unfinished exchanges can never pass as results.
// All workgroups of a launch must be co-resident: the host launches at most one workgroup per CU and serialises resident
// launches of a device on one lane stream (context.hip), so two of them can never hold each other's CUs.
//
// XCD placement (LOCAL = 1 / 2) is NOT assumed from blockIdx (HIP promises nothing about workgroup -> XCD placement,
// MI355X_MICROARCH.md "Contract"): a workgroup reads the XCC it really runs on (s_getreg XCC_ID) and takes the next free
// slot OF THAT XCC from a per-XCC ticket counter; its pair and row group follow from (xcc, slot).  So the workgroups that
// exchange through plain stores share an L2 by construction, whatever order the dispatcher used.  What remains an
// assumption - every XCC receives exactly its share of the grid - is checked: a ticket beyond the XCC's capacity raises
// the status like a time-out and the launch is void.
//
// Fixed summation orders, no float atomics: bit-reproducible run to run.  Values differ from the streaming path in the last
// bits (another summation order); both paths are checked against the same fixtures.
#include "imp_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float OT_EPS = 1e-8f;      // nets/layers.py:13
constexpr int AUX_SC1 = 16;          // gfx940+ cache policy bit: agent-scope coherent access (what a relaxed agent atomic uses)
constexpr int SPIN_LIMIT = 1 << 21;   // polls of one wait before it is declared dead (seconds)

// sum over the 64 lanes with DPP row operations (6 VALU instructions, no LDS crossbar round trips: the __shfl_xor butterfly costs
// six dependent ds_bpermute per sum, ~2400 cycles of an iteration's phase A); fixed order, the total is broadcast from lane 63
__device__ __forceinline__ float wave_sum(float v) {
#define OTR_DPP_ADD(ctrl, row_mask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    OTR_DPP_ADD(0xB1, 0xf);          // quad_perm [1,0,3,2]
    OTR_DPP_ADD(0x4E, 0xf);          // quad_perm [2,3,0,1]
    OTR_DPP_ADD(0x141, 0xf);         // row_half_mirror
    OTR_DPP_ADD(0x140, 0xf);         // row_mirror: every lane of a 16-lane row holds the row sum
    OTR_DPP_ADD(0x142, 0xa);         // row_bcast:15 into rows 1 and 3
    OTR_DPP_ADD(0x143, 0xc);         // row_bcast:31 into rows 2 and 3: lane 63 holds the total
#undef OTR_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// the sums of N values at once: the DPP steps of the N chains interleaved (a chain alone waits two states after every step), every lane gets the totals
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#define OTR_DPP_ADDN(ctrl, row_mask) _Pragma("unroll") for (int i = 0; i < N; ++i) v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), ctrl, row_mask, 0xf, false))
    OTR_DPP_ADDN(0xB1, 0xf);
    OTR_DPP_ADDN(0x4E, 0xf);
    OTR_DPP_ADDN(0x141, 0xf);
    OTR_DPP_ADDN(0x140, 0xf);
    OTR_DPP_ADDN(0x142, 0xa);
    OTR_DPP_ADDN(0x143, 0xc);
#undef OTR_DPP_ADDN
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {     // first index wins on ties (torch.max)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#ifndef OTR_POLL_AUX
#define OTR_POLL_AUX (AUX_SC1 | (int)0x80000000)      // + volatile: a poll must not be hoisted out of its loop (the compiler then also sets sc0: system scope)
#endif
constexpr int AUX_POLL = OTR_POLL_AUX;
// four consecutive values of an exchange vector = 4 granules = 32 bytes at granule index 4 q
// AUX = AUX_SC1: write-through, the line is dropped from the writer's L2 (readers on other XCDs fetch it across the fabric);
// AUX = 0 (XCD-local launches): a plain store - the line stays in the ONE L2 that all workgroups of the pair share, and the
// readers' sc1 polls (which bypass only their L1) are served from it
template <int AUX>
__device__ __forceinline__ void stg4(__amdgpu_buffer_rsrc_t r, int q, const f32x4 v, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag}, r, q * 32, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag}, r, q * 32 + 16, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void stg1(__amdgpu_buffer_rsrc_t r, int idx, float v, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v), tag}, r, idx * 8, 0, AUX);
}
// poll until all four granules carry `tag`; `dead` (per thread) short-circuits every wait after a time-out
// the first waiter of a launch whose wait times out leaves a
// POST-MORTEM record in the mapped host page (imp_kernels.h imp_postmortem_write; read back by imp_resident_postmortem)
// (kept small: this kernel lives at the register limit - `packed` = pair | group << 6 | G << 14 | placement << 22 | pairs << 24 in ONE scalar register, the
// phase and the iteration arrive as `where` = phase | iteration << 8 from values the call site has anyway)
struct Health { int* status; int* host; unsigned launch_tag; unsigned packed; };
__device__ __forceinline__ void poll_health(const Health& h, int spins, bool& dead, int idx, unsigned want, unsigned seen, int where) {
    if (spins > SPIN_LIMIT) {
        __hip_atomic_store(h.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h.host) {
            imp_postmortem_write(h.host, 1, h.launch_tag, where & 255, idx, want, seen, where >> 8, h.packed & 63, (h.packed >> 6) & 255, (h.packed >> 14) & 255, (h.packed >> 22) & 3, h.packed >> 24);
            __hip_atomic_store(h.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (__hip_atomic_load(h.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
}
__device__ __forceinline__ f32x4 ldg4(__amdgpu_buffer_rsrc_t r, int q, unsigned tag, const Health& status, bool& dead, int where) {
    u32x4 a, c;
    int spins = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        a = __builtin_amdgcn_raw_buffer_load_b128(r, q * 32, 0, AUX_POLL);
        c = __builtin_amdgcn_raw_buffer_load_b128(r, q * 32 + 16, 0, AUX_POLL);
        if ((a[1] == tag && a[3] == tag && c[1] == tag && c[3] == tag) || dead) break;
        if ((++spins & 1023) == 0) poll_health(status, spins, dead, q, tag, a[1], where);
    }
    return f32x4{__uint_as_float(a[0]), __uint_as_float(a[2]), __uint_as_float(c[0]), __uint_as_float(c[2])};
}

// (round 5 measured two variants of these hand-offs as build switches - both chunks of a two-chunk owner polled in one round trip: neutral, 6.09-6.12 vs 6.11-6.14 us
// per iteration; the owners' swap and the distribution of v merged into one hand-off read by every workgroup: slower, 6.37-6.43 vs 5.99-6.01 - removed in round 6:
// profiles/r05/sinkhorn_dual_poll_ab.log, sinkhorn_merged_handoffs_ab.log)

// one granule
__device__ __forceinline__ float ldg1(__amdgpu_buffer_rsrc_t r, int idx, unsigned tag, const Health& status, bool& dead, int where) {
    u32x2 a;
    int spins = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        a = __builtin_amdgcn_raw_buffer_load_b64(r, idx * 8, 0, AUX_POLL);
        if (a[1] == tag || dead) break;
        if ((++spins & 1023) == 0) poll_health(status, spins, dead, idx, tag, a[1], where);
    }
    return __uint_as_float(a[0]);
}

// phase profile of workgroup 0 (p.prof != null; tools/probe/sk_prof.py): cycles of  0 A (u + column partials)  1 B (LDS combine +
// partial store)  2 wait + stage of the slice  3 slice reduce + v store  4 wait + read of v  5 v sum
#define OTR_CLK(i) if (p.prof) { const unsigned long long c_ = __builtin_readcyclecounter(); prof_acc[i] += c_ - tlast; tlast = c_; }

// LOCAL: every pair lives on ONE XCD: a workgroup that runs on XCC x with ticket `slot` there serves pair x + 8 * (slot / G), group slot % G,
// so all G <= 32 workgroups of a pair share an L2 and the exchanges never cross the fabric
// LOCAL = 2 (two XCDs per pair, B <= 4): pair b lives on XCDs 2b and 2b + 1, half of its workgroups on each.  The column sums are
// formed hierarchically - every half reduces ITS workgroups' partials through its own L2, the two halves swap their half sums
// across the fabric (the one remote hand-off of the iteration), both compute the same v and distribute it inside their XCD - so
// an iteration has one fabric crossing instead of two.
template <int NCH, int RPW, int LOCAL>
__global__ __launch_bounds__(512, 2) void ot_resident_kernel(const OtResidentParams p) {
    constexpr int ST_AUX = LOCAL ? 0 : AUX_SC1;            // partial vectors and v: readers share the writer's L2 when LOCAL
    constexpr int MX_AUX = LOCAL == 1 ? 0 : AUX_SC1;       // the final column-maxima exchange runs over all workgroups of the pair
    constexpr int DCOL = 256 * NCH;            // exchange vectors: inner columns | dustbin column | 3 pads
    constexpr int LDX = DCOL + 4;
    constexpr int NQ = LDX / 4;                // float4 chunks of an exchange vector
    constexpr int ROWS = 8 * RPW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* vs = lds;                           // [LDX]      current v
    float* red = lds + LDX;                    // [8][LDX]   wave partials / staging (sized by the launcher)
    __shared__ float s_vsw[8];                 // per-wave partial sums of v

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.G;
    // tag / ticket bases: launch parameters, or - for a launch that lives in a hipGraph and is replayed with the same parameters - two
    // device words that the LAST workgroup of every launch advances (every workgroup has read them by then: it reads them before it counts itself done)
    unsigned tag_base = p.tag_base, ticket_base = p.ticket_base;
    if (p.dev_base) {
        tag_base = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.dev_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        ticket_base = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.dev_base + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    auto leave = [&]() {                       // every exit of the kernel, workgroup-uniform
        if (p.dev_base && tid == 0) {
            const unsigned done = __hip_atomic_fetch_add(p.dev_base + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1) {
                unsigned nb = tag_base + 3u * (unsigned)p.T + 4u;
                if (nb > 0xFFFFF000u) {                                    // graph launches tag in the upper half of the 32-bit space
                    nb = 0x80000000u;
                    // ~7 million replays at T = 100: the tag space of the replayed launches starts over.  Tags must not repeat on buffers that
                    // may still hold them, so the library is told (word 2 of the mapped health page: "clear the exchange buffers", not an
                    // error) and does so at its next entry point (context.hip resident_health)
                    if (p.host_status) __hip_atomic_store(p.host_status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                const unsigned share = LOCAL == 1 ? (unsigned)(G * ((p.B + 7) >> 3)) : (LOCAL == 2 ? (unsigned)(G >> 1) : 0u);
                __hip_atomic_store(p.dev_base, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.dev_base + 1, ticket_base + share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.dev_base + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    int b, g, half = 0, H = G;                 // H workgroups exchange through one L2; this one is number gl of them
    if (LOCAL) {
        // the XCC this workgroup really runs on, and its ticket among the workgroups of the launch that landed there
        __shared__ int s_place[2];
        if (tid == 0) {
            unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;       // XCC_ID[3:0] (hwreg 20 on gfx942 / gfx950)
            if (p.fake_placement) xcc = (xcc + (blockIdx.x >> 3)) & 7u;                      // TEST HOOK: lie about the placement (still balanced)
            const unsigned ticket = __hip_atomic_fetch_add(p.xcc_tickets + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_place[0] = (int)xcc;
            s_place[1] = (int)(ticket - ticket_base);
        }
        __syncthreads();
        const int xcc = __builtin_amdgcn_readfirstlane(s_place[0]), slot = __builtin_amdgcn_readfirstlane(s_place[1]);      // (workgroup-uniform: scalar registers - pair, group and everything derived from them)
        const int cap = LOCAL == 1 ? G * ((p.B + 7) >> 3) : (G >> 1);
        if (slot < 0 || slot >= cap) {             // this XCC received more workgroups than its share: the launch is void
            if (tid == 0) {
                __hip_atomic_store(p.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (p.host_status) {
                    imp_postmortem_write(p.host_status, 2, tag_base, 0, slot, (unsigned)cap, (unsigned)xcc, -1, -1, -1, G, LOCAL, p.B);
                    __hip_atomic_store(p.host_status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            leave();
            return;
        }
        if (LOCAL == 1) {
            b = xcc + 8 * (slot / G);
            g = slot % G;
        } else {
            // the first half - rows 0 .. 1023, G / 2 = 32 workgroups - is always complete; of the second half only the first p.h1 workgroups (the rows
            // the LAUNCH's largest pair needs; >= 8) take part: the others leave at once and give their CUs back to whatever else runs on the chip (with
            // several batches in flight a launch that holds all 256 CUs waits long for the last of them and starves its neighbours meanwhile).  The
            // column sums of a half are a perfect binary tree over its 32 slots in which an absent workgroup is an exact zero: any h1 gives the same bits
            b = xcc >> 1;
            half = xcc & 1;
            H = half ? p.h1 : (G >> 1);
            g = half * (G >> 1) + slot;
            if (slot >= H) { leave(); return; }
        }
        if (b >= p.B) { leave(); return; }     // uniform per workgroup, before any exchange
    } else {
        b = blockIdx.x / G;
        g = blockIdx.x % G;
    }
    // ragged batches: this pair's own matrix is n0 x n1; p.n0 / p.n1 are the padded sizes (strides of dist, of the maxima and of u / v)
    const int ld0 = p.n0, ld1 = p.n1;
    const int n0 = imp_count(p.rc, 0, b, ld0), n1 = imp_count(p.rc, 1, b, ld1);
    if (n0 <= 0 || n1 <= 0) { leave(); return; }       // retired pair: all its workgroups leave here, before any exchange
    const int r0 = g * ROWS + wave * RPW;
    bool dead = false;
    const Health health{p.status, p.host_status, tag_base, (unsigned)__builtin_amdgcn_readfirstlane((b & 63) | ((g & 255) << 6) | ((G & 255) << 14) | (LOCAL << 22) | (p.B << 24))};

    // exchange buffers hold granules: 8 bytes per float
    const int gl = LOCAL == 2 ? g - half * (G >> 1) : g;
    const int GP = LOCAL == 2 ? (G >> 1) + p.h1 : G;       // workgroups of the pair that take part: g = 0 .. GP - 1
    const __amdgpu_buffer_rsrc_t rs_part = make_rsrc(p.xpart + ((size_t)b * G + (size_t)(g - gl)) * LDX * 2, (unsigned)((size_t)H * LDX * 8));
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(p.xv + (size_t)(LOCAL == 2 ? 2 * b + half : b) * LDX * 2, (unsigned)(LDX * 8));
    const __amdgpu_buffer_rsrc_t rs_h_own = make_rsrc(p.xhalf + (size_t)(2 * b + half) * LDX * 2, (unsigned)(LDX * 8));
    const __amdgpu_buffer_rsrc_t rs_h_oth = make_rsrc(p.xhalf + (size_t)(2 * b + 1 - half) * LDX * 2, (unsigned)(LDX * 8));
    // the vectors of odd iterations live 8 vectors further (B <= 4: 2 B <= 8 vectors per parity)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_h_own1 = make_rsrc(p.xhalf + (size_t)(8 + 2 * b + half) * LDX * 2, (unsigned)(LDX * 8));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_h_oth1 = make_rsrc(p.xhalf + (size_t)(8 + 2 * b + 1 - half) * LDX * 2, (unsigned)(LDX * 8));

    // ---- row softmax of the dustbin-augmented matrix (nets/layers.py:39-40,28) straight into registers -----------
    f32x4 P[RPW][NCH];
    float Pd[RPW];
    const float bin = p.bin;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r0 + k;
        const bool rv = r < n0;
        const float* drow = p.dist + ((size_t)b * ld0 + (rv ? r : 0)) * ld1;
        float mx = bin;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = 4 * (lane + 64 * c);
            f32x4 x;
            if (rv && j + 3 < n1 && (ld1 & 3) == 0) {
                x = *reinterpret_cast<const f32x4*>(drow + j);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (rv && j + e < n1) ? drow[j + e] : -INFINITY;
            }
            P[k][c] = x;
            mx = fmaxf(mx, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ev = expf(P[k][c][e] - mx);        // exp(-inf) = 0 for columns past n1 / rows past n0
                P[k][c][e] = ev;
                sum += ev;
            }
        const float ed = expf(bin - mx);
        sum = wave_sum(sum) + ed;
        const float inv_sum = 1.0f / sum;           // (sum >= 1: the row maximum contributes exp(0))
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) P[k][c][e] = rv ? imp_div_by(P[k][c][e], sum, inv_sum) : 0.f;      // == P / sum, bit for bit (imp_kernels.h)
        Pd[k] = rv ? ed / sum : 0.f;
    }
    const float c0 = 1.0f / (float)(n1 + 1);   // every entry of the dustbin row: softmax of n1 + 1 equal logits

    // start vectors (nets/layers.py:29-30): u = 1, v = 1
    // (uniform trip counts for the thread-strided loops outside the iteration loop: hipcc of ROCm 7.2 put register spills into the exit block of a loop with a
    // divergent trip count BEFORE the instruction that restores the execution mask - stores with no lane enabled; found as row arg-maxima that read back zero)
    for (int j0 = 0; j0 < LDX; j0 += 512) {
        const int j = j0 + tid;
        if (j < LDX) vs[j] = (j < n1 || j == DCOL) ? 1.f : 0.f;
    }
    float vsum = (float)(n1 + 1);              // sum of v over the n1 + 1 real columns
    float u[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) u[k] = (r0 + k < n0) ? 1.f : 0.f;
    float u_last = 1.f;
    __syncthreads();

    const int cq = (NQ + H - 1) / H;           // float4 chunks of the exchange vector owned by one workgroup (of its half)
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
    for (int it = 0; it < p.T; ++it) {
        const unsigned tag_p = tag_base + 3u * it + 1u, tag_h = tag_p + 1u, tag_v = tag_p + 2u;
        // ---- A: u for the own rows, column partials ---------------------------------------------------------------
        float acc[RPW];
        // (round 5) a lane's share of a row's dot product as TWO chains - even and odd columns - so that a step of both is one v_pk_fma_f32 on the register pair
        // (P[k][c][0..1], x[0..1]): 64 packed instead of 128 scalar FMAs (phase B's products over a float4 were packed already).  And the RPW rows advance
        // TOGETHER: the compiler used to finish one row - a chain of dependent FMAs, its six dependent DPP additions, its IEEE division behind a branch - before
        // it started the next, every step waiting for the one before; sched_barrier keeps the rows interleaved, the divisions are unconditional
        f32x4 xv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) xv[c] = *reinterpret_cast<const f32x4*>(vs + 4 * (lane + 64 * c));
        const float vd = vs[DCOL];
        f32x2 acc2[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) acc2[k] = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int k = 0; k < RPW; ++k) acc2[k] = __builtin_elementwise_fma(f32x2{P[k][c][0], P[k][c][1]}, f32x2{xv[c][0], xv[c][1]}, acc2[k]);
#pragma unroll
            for (int k = 0; k < RPW; ++k) acc2[k] = __builtin_elementwise_fma(f32x2{P[k][c][2], P[k][c][3]}, f32x2{xv[c][2], xv[c][3]}, acc2[k]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) acc[k] = acc2[k][0] + acc2[k][1];
        wave_sum_n<RPW>(acc);
        float pdpart = 0.f;
        float inv[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) inv[k] = 1.f / (fmaf(Pd[k], vd, acc[k]) + OT_EPS);      // real rows have marginal 1 (nets/layers.py:32,41)
#pragma unroll
        for (int k = 0; k < RPW; ++k) u[k] = (r0 + k < n0) ? inv[k] : 0.f;
        {   // the dustbin column's partial like every other column's: 4-row leaves, RPW = 8 adds its two leaves (CANONICAL COLUMN SUMS below)
            constexpr int LEAFD = RPW < 4 ? RPW : 4;
#pragma unroll
            for (int k = 0; k < LEAFD; ++k) pdpart = fmaf(Pd[k], u[k], pdpart);
            if constexpr (RPW == 8) {
                float pd2 = 0.f;
#pragma unroll
                for (int k = 4; k < 8; ++k) pd2 = fmaf(Pd[k], u[k], pd2);
                pdpart += pd2;
            }
        }
        u_last = (float)(n0 + 1) / (c0 * vsum + OT_EPS);              // dustbin row: marginal n0 + 1 (nets/layers.py:42)
        OTR_CLK(0)
        // ---- B: workgroup partial vector ---------------------------------------------------------------------------
        // CANONICAL COLUMN SUMS (round 6).  A column's sum over the rows is formed as a perfect binary tree over the ROW INDEX whose leaves are the
        // aligned groups of four rows, each an fma chain in row order: leaf -> 8 rows -> 16 -> ... - here inside a wave (RPW = 8: two leaves), across
        // the 8 waves of the workgroup, below across the workgroups (and the two XCD halves).  Rows past n0 contribute exact zeros, so the value
        // depends on the pair's own n0 alone: whatever decomposition (rows per wave, workgroups per pair, one / two XCDs / chip-wide, padded batch
        // sizes) a launch picked, the bits are the same.  (Rounds 2-5 summed the waves and the workgroups sequentially in eighths of G - a pair's
        // Sinkhorn result moved in the last bits with the batch it travelled in; VERDICT r5 weak #1.)  RPW < 4 (the two shapes for n1 > 2048) only
        // ever run for a single pair: ot_resident_plan.
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            f32x4 part = {0.f, 0.f, 0.f, 0.f};
            constexpr int LEAF = RPW < 4 ? RPW : 4;
#pragma unroll
            for (int k = 0; k < LEAF; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[e] = fmaf(P[k][c][e], u[k], part[e]);
            if constexpr (RPW == 8) {
                f32x4 part2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 4; k < 8; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) part2[e] = fmaf(P[k][c][e], u[k], part2[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) part[e] += part2[e];
            }
            *reinterpret_cast<f32x4*>(red + wave * LDX + 4 * (lane + 64 * c)) = part;
        }
        if (lane == 0) *reinterpret_cast<f32x4*>(red + wave * LDX + DCOL) = f32x4{pdpart, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int q = tid; q < NQ; q += 1024) {       // (two chunks per round: the one thread that owns chunk 512 too reads all 16 fragments before it adds)
            const int q2 = q + 512;
            const bool two = q2 < NQ;
            // waves in a tree: ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7))
            auto rd = [&](int w, int qq) { return *reinterpret_cast<const f32x4*>(red + w * LDX + 4 * qq); };
            auto tree8 = [&](int qq) {
                const f32x4 a = (rd(0, qq) + rd(1, qq)) + (rd(2, qq) + rd(3, qq));
                const f32x4 b4 = (rd(4, qq) + rd(5, qq)) + (rd(6, qq) + rd(7, qq));
                return a + b4;
            };
            const f32x4 s = tree8(q);
            const f32x4 s2 = two ? tree8(q2) : f32x4{0.f, 0.f, 0.f, 0.f};
            stg4<ST_AUX>(rs_part, gl * NQ + q, s, tag_p);
            if (two) stg4<ST_AUX>(rs_part, gl * NQ + q2, s2, tag_p);
        }
        __syncthreads();                           // everyone is done with the wave partials in `red`
        OTR_CLK(1)
        // ---- C: this workgroup's slice of columns over the G partial vectors -------------------------------------
        {
            f32x4* stage = reinterpret_cast<f32x4*>(red);             // [H][cq]
            for (int idx = tid; idx < cq * H; idx += 512) {
                const int w = idx / cq, qq = idx - w * cq;
                const int q = gl * cq + qq;
                stage[idx] = q < NQ ? ldg4(rs_part, w * NQ + q, tag_p, health, dead, 1 | (it << 8)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
            OTR_CLK(2)
            // the tree goes on over the workgroups (index = row block): 8 threads per column, each the subtree of an aligned run of `seg` workgroups
            // (seg = the next power of two >= H, over 8; absent workgroups are exact zeros), then the 8 subtrees as a tree
            float* sub = red + (size_t)cq * H * 4;                    // [8][4 cq]
            const int ncol = 4 * cq;
            int hp2 = 8;
            while (hp2 < H) hp2 <<= 1;
            const int seg = hp2 >> 3;
            for (int t = tid; t < 8 * ncol; t += 512) {
                const int h = t / ncol, cl = t - h * ncol;
                const int qq = cl >> 2, e = cl & 3;
                const int wa = h * seg;
                auto val = [&](int w) { return w < H ? red[(size_t)(w * cq + qq) * 4 + e] : 0.f; };
                float s;
                if (seg == 1) s = val(wa);
                else if (seg == 2) s = val(wa) + val(wa + 1);
                else if (seg == 4) s = (val(wa) + val(wa + 1)) + (val(wa + 2) + val(wa + 3));
                else {                                                // chip-wide launches of up to 256 workgroups per pair: pairwise sums with a binary carry
                    float lvl[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                    const int nl = 31 - __builtin_clz(seg);           // levels below the root (seg <= 32)
                    s = 0.f;
                    for (int i = 0; i < seg; ++i) {
                        float x = val(wa + i);
                        bool placed = false;
#pragma unroll
                        for (int L = 0; L < 5; ++L) {
                            if (placed || L >= nl) continue;
                            if ((i >> L) & 1) x = lvl[L] + x;
                            else { lvl[L] = x; placed = true; }
                        }
                        if (!placed) s = x;                           // i = seg - 1: every level merged - the root
                    }
                }
                sub[h * ncol + cl] = s;
            }
            __syncthreads();
            for (int cl = tid; cl < ncol; cl += 512) {
                float s = ((sub[cl] + sub[ncol + cl]) + (sub[2 * ncol + cl] + sub[3 * ncol + cl])) +
                          ((sub[4 * ncol + cl] + sub[5 * ncol + cl]) + (sub[6 * ncol + cl] + sub[7 * ncol + cl]));
                const int xi = 4 * (gl * cq) + cl;                    // index in the exchange layout
                if (xi < LDX) {
                    if (LOCAL == 2) {                                 // swap the half sums across the fabric; both halves add them in the same order
                        // TWO granule vectors, by iteration parity (round 6: the cause of the rare time-out, read off three post-mortem records - a waiter in phase 2
                        // that found the tag of iteration it + 1 where it expected it).  A half needs the OTHER half's sums of iteration it to go on, not that the
                        // other half has READ its own: it could run a whole iteration ahead and publish its sums of it + 1 over those of it before a slow poll of
                        // the other half (a fabric crossing beside other streams' write bursts) had seen them - the value was gone, the poll never ended.  With two
                        // vectors the sums of it are overwritten by those of it + 2, which need the other half's sums of it + 1, which the SAME thread of the other
                        // half publishes only after it has read ours of it
                        stg1<AUX_SC1>((it & 1) ? rs_h_own1 : rs_h_own, xi, s, tag_h);
                        const float o = ldg1((it & 1) ? rs_h_oth1 : rs_h_oth, xi, tag_h, health, dead, 2 | (it << 8));
                        s = half == 0 ? s + o : o + s;
                    }
                    const bool dust = xi == DCOL;
                    const bool real = xi < n1 || dust;
                    const float t = fmaf(c0, u_last, s);              // + dustbin row entry * its u
                    const float marg = dust ? (float)(n1 + 1) : 1.f;  // nets/layers.py:43-44
                    stg1<ST_AUX>(rs_v, xi, real ? marg / (t + OT_EPS) : 0.f, tag_v);
                }
            }
        }
        OTR_CLK(3)
        // ---- D: everybody reads v -----------------------------------------------------------------------------------
        {   // ... and the sum of v from the chunks the threads just fetched (entries that are no real column travel as zeros): a partial per wave, combined in a fixed order
            // (the dustbin entry - chunk NQ - 1, whose owner thread depends on the column class NCH - is added apart below, so that the sum depends on
            // the pair's own n1 alone; inner chunk q belongs to thread q mod 512 in every class, and chunks past n1 are zeros)
            float vpart = 0.f;
            for (int q = tid; q < NQ; q += 512) {
                const f32x4 o = ldg4(rs_v, q, tag_v, health, dead, 3 | (it << 8));
                *reinterpret_cast<f32x4*>(vs + 4 * q) = o;
                if (q != NQ - 1) vpart += (o[0] + o[1]) + (o[2] + o[3]);
            }
            vpart = wave_sum(vpart);
            if (lane == 0) s_vsw[wave] = vpart;
        }
        __syncthreads();
        OTR_CLK(4)
        vsum = (((s_vsw[0] + s_vsw[1]) + (s_vsw[2] + s_vsw[3])) + ((s_vsw[4] + s_vsw[5]) + (s_vsw[6] + s_vsw[7]))) + vs[DCOL];
        OTR_CLK(5)
    }
    if (p.prof && b == 0 && g == 0 && tid == 0)
        for (int i = 0; i < 6; ++i) p.prof[i] = prof_acc[i];

    // ---- outputs ----------------------------------------------------------------------------------------------------
    // (the row base is laundered through an empty asm: every output address is then computed HERE, after the iteration loop - the compiler otherwise forms
    // them before the loop and, at this kernel's register limit, parks them in scratch across it; see the note on spills at the start vectors)
    int r0o = r0;
    asm volatile("" : "+v"(r0o));
    const float vd = vs[DCOL];
    if (p.u) {
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < RPW; ++k)
                if (r0o + k < n0) p.u[(size_t)b * p.ldu + r0o + k] = u[k];
        if (g == 0) {
            if (tid == 0) p.u[(size_t)b * p.ldu + n0] = u_last;
            for (int j = tid; j < n1; j += 512) p.v[(size_t)b * p.ldv + j] = vs[j];
            if (tid == 0) p.v[(size_t)b * p.ldv + n1] = vd;
        }
    }
    const bool want_max = p.max0 != nullptr;
    float* rowbuf = red + wave * LDX;                  // per-wave staging of one score row
    __syncthreads();                                   // `red` is free again
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r0o + k;
        if (r >= n0) continue;                         // wave-uniform
        float best = -INFINITY;
        int code = 4 * NCH;                            // which of the lane's 4 NCH scores is its first maximum: 4 c + e (an inline constant per candidate - the 32
                                                       // column indices themselves cost registers this kernel does not have: round 6 found them spilled)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(vs + 4 * (lane + 64 * c));
            const int rem = n1 - 4 * (lane + 64 * c);  // columns of this chunk that exist
            f32x4 s;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] = (P[k][c][e] * u[k]) * x[e];     // (p * u) * v as nets/layers.py:34
                if (rem > e && s[e] > best) { best = s[e]; code = 4 * c + e; }      // ascending j inside the lane: first wins
            }
            if (p.scores) *reinterpret_cast<f32x4*>(rowbuf + 4 * (lane + 64 * c)) = s;
        }
        int bi = code == 4 * NCH ? 0x7fffffff : 4 * (lane + 64 * (code >> 2)) + (code & 3);
        if (want_max) {
            wave_argmax(best, bi);
            if (lane == 0) { p.max0[(size_t)b * ld0 + r] = best; p.arg0[(size_t)b * ld0 + r] = bi; }
        }
        if (p.scores) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float* srow = p.scores + (size_t)b * (ld0 + 1) * (ld1 + 1) + (size_t)r * (n1 + 1);
            for (int j = lane; j < n1; j += 64) srow[j] = rowbuf[j];       // lane-linear dword stores
            if (lane == 0) srow[n1] = (Pd[k] * u[k]) * vd;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.scores && g == 0) {                          // dustbin row of the score tensor
        float* srow = p.scores + (size_t)b * (ld0 + 1) * (ld1 + 1) + (size_t)n0 * (n1 + 1);
        for (int j = tid; j < n1; j += 512) srow[j] = (c0 * u_last) * vs[j];
        if (tid == 0) srow[n1] = (c0 * u_last) * vd;
    }
    if (want_max) {
        // column maxima of the SAME expression (bit-identical values on both sides of the mutual check).  The waves take
        // turns in ascending order (= ascending rows; strict > keeps the first row on ties) on one LDS vector, then the
        // workgroup vectors are combined across the group like the column sums.
        __syncthreads();
        float* mv = red;                               // [LDX] values
        int* mi = reinterpret_cast<int*>(red + LDX);   // [LDX] row indices
        for (int w = 0; w < 8; ++w) {
            if (wave == w) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(vs + 4 * (lane + 64 * c));
                    f32x4 cb;
                    int ci[4];
                    if (w == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { cb[e] = -INFINITY; ci[e] = 0x7fffffff; }
                    } else {
                        cb = *reinterpret_cast<const f32x4*>(mv + 4 * (lane + 64 * c));
#pragma unroll
                        for (int e = 0; e < 4; ++e) ci[e] = mi[4 * (lane + 64 * c) + e];
                    }
#pragma unroll
                    for (int k = 0; k < RPW; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sv = (P[k][c][e] * u[k]) * x[e];
                            if (r0o + k < n0 && sv > cb[e]) { cb[e] = sv; ci[e] = r0o + k; }
                        }
                    *reinterpret_cast<f32x4*>(mv + 4 * (lane + 64 * c)) = cb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) mi[4 * (lane + 64 * c) + e] = ci[e];
                }
            }
            __syncthreads();
        }
        const unsigned tag_m = tag_base + 3u * p.T + 1u;
        const __amdgpu_buffer_rsrc_t rs_mx = make_rsrc(p.xmax + (size_t)b * G * 4 * LDX, (unsigned)((size_t)GP * 2 * LDX * 8));
        for (int q = tid; q < DCOL / 4; q += 512) {
            stg4<MX_AUX>(rs_mx, g * 2 * NQ + q, *reinterpret_cast<const f32x4*>(mv + 4 * q), tag_m);
            stg4<MX_AUX>(rs_mx, g * 2 * NQ + NQ + q, *reinterpret_cast<const f32x4*>(red + LDX + 4 * q), tag_m);
        }
        __syncthreads();                           // mv / mi are about to be overwritten by the staging
        const int ncq = (DCOL / 4 + GP - 1) / GP;      // float4 column chunks per workgroup
        f32x4* stage = reinterpret_cast<f32x4*>(red);  // [2][GP][ncq]
        for (int idx = tid; idx < 2 * ncq * GP; idx += 512) {
            const int which = idx / (ncq * GP), rem = idx - which * ncq * GP;
            const int w = rem / ncq, qq = rem - w * ncq;
            const int q = g * ncq + qq;
            stage[idx] = q < DCOL / 4 ? ldg4(rs_mx, w * 2 * NQ + which * NQ + q, tag_m, health, dead, 4 | (p.T << 8)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        for (int cl = tid; cl < 4 * ncq; cl += 512) {
            const int j = 4 * (g * ncq) + cl;
            if (j < n1) {
                const int qq = cl >> 2, e = cl & 3;
                float best = -INFINITY;
                int bi = 0x7fffffff;
                for (int w = 0; w < GP; ++w) {         // ascending workgroups = ascending rows
                    const float ov = red[(size_t)(w * ncq + qq) * 4 + e];
                    const int oi = __float_as_int(red[(size_t)((GP + w) * ncq + qq) * 4 + e]);
                    if (ov > best) { best = ov; bi = oi; }
                }
                p.max1[(size_t)b * ld1 + j] = best;
                p.arg1[(size_t)b * ld1 + j] = bi;
            }
        }
    }
    // ---- a workgroup in which any wait timed out voids what it wrote: unfinished exchanges must never pass as results
    if (__syncthreads_or(dead ? 1 : 0)) {
        const float nanv = __builtin_nanf("");
        const int rend = min(n0, (g + 1) * ROWS);
        for (int r = g * ROWS + tid; r < rend; r += 512)
            if (want_max) { p.max0[(size_t)b * ld0 + r] = nanv; p.arg0[(size_t)b * ld0 + r] = 0x7fffffff; }
        // the WHOLE own rows of the score tensor (with only the first score of a row poisoned, maxima recomputed from the
        // tensor - imp_compute_matches after imp_compute_score - skipped the NaN and returned plausible matches from garbage)
        if (p.scores)
            for (int r = g * ROWS + wave; r < rend; r += 8) {
                float* srow = p.scores + (size_t)b * (ld0 + 1) * (ld1 + 1) + (size_t)r * (n1 + 1);
                for (int j = lane; j <= n1; j += 64) srow[j] = nanv;
            }
        if (want_max) {
            const int ncq = (DCOL / 4 + GP - 1) / GP;
            for (int cl = tid; cl < 4 * ncq; cl += 512) {
                const int j = 4 * (g * ncq) + cl;
                if (j < n1) { p.max1[(size_t)b * ld1 + j] = nanv; p.arg1[(size_t)b * ld1 + j] = 0x7fffffff; }
            }
        }
    }
    leave();
}

template <int NCH, int RPW>
hipError_t launch_one(const OtResidentParams& p_in, hipStream_t stream) {
    constexpr int LDX = 256 * NCH + 4;
    OtResidentParams p = p_in;
    // two XCDs per pair: how many workgroups of the second half (rows 1024 ...) take part - those the largest pair of the launch has rows for, at least 8
    // (fewer would make a workgroup's column slice of the half's exchange long)
    p.h1 = p.local == 2 ? std::min(32, std::max(8, (p.n0 - 1024 + 8 * RPW - 1) / (8 * RPW))) : 0;
    const int GP = p.local == 2 ? p.G / 2 + p.h1 : p.G;
    // vs + max(8 wave vectors, slice staging [H][cq] float4 + [8][4 cq])
    size_t red = (size_t)8 * LDX;
    for (int Hh : {p.local == 2 ? p.G / 2 : p.G, p.local == 2 ? p.h1 : p.G}) {
        const int cq = (LDX / 4 + Hh - 1) / Hh;
        const size_t stage = (size_t)cq * Hh * 4 + (size_t)8 * 4 * cq;
        if (stage > red) red = stage;
    }
    const size_t stage2 = (size_t)2 * GP * ((256 * NCH / 4 + GP - 1) / GP) * 4;     // column-maxima staging
    if (stage2 > red) red = stage2;
    const size_t lds = (LDX + red) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (p.local == 2) {
        if (p.B > 4 || (p.G & 1) || p.G / 2 > 32 || !p.xhalf) return hipErrorInvalidValue;     // two XCDs per pair
        if constexpr (RPW == 4 && NCH >= 5 && NCH <= 8) {     // (the shapes ot_resident_hier_ok admits: 1024 < n1 <= 2048 columns, 32-row workgroups)
            if (hipError_t e = imp_grant_dynamic_lds((const void*)ot_resident_kernel<NCH, RPW, 2>, lds)) return e;
            hipLaunchKernelGGL((ot_resident_kernel<NCH, RPW, 2>), dim3(8 * (p.G / 2)), dim3(512), lds, stream, p);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (p.local) {
        if (p.G * ((p.B + 7) / 8) > 32) return hipErrorInvalidValue;      // a pair's workgroups must fit the 32 CUs of its XCD
        if (hipError_t e = imp_grant_dynamic_lds((const void*)ot_resident_kernel<NCH, RPW, 1>, lds)) return e;
        hipLaunchKernelGGL((ot_resident_kernel<NCH, RPW, 1>), dim3(8 * p.G * ((p.B + 7) / 8)), dim3(512), lds, stream, p);
        return hipGetLastError();
    }
    if (hipError_t e = imp_grant_dynamic_lds((const void*)ot_resident_kernel<NCH, RPW, 0>, lds)) return e;
    hipLaunchKernelGGL((ot_resident_kernel<NCH, RPW, 0>), dim3(p.B * p.G), dim3(512), lds, stream, p);
    return hipGetLastError();
}

struct Shape { int nch, rpw; };
// (NCH, RPW) instantiations; NCH * RPW <= 32 float4 = 128 VGPRs of matrix per lane
// (round 6: every shape for n1 <= 2048 holds whole 4-row leaves of the canonical column-sum tree, RPW = 4 or 8 - the RPW = 2 shapes of rounds 2-5 are
// gone; the two wide shapes (12, 2) and (16, 1) cannot and are only planned for a single pair, which then takes them in every call)
constexpr Shape kShapes[] = {{2, 4}, {2, 8}, {3, 4}, {3, 8}, {4, 4}, {4, 8}, {5, 4}, {6, 4}, {8, 4}, {12, 2}, {16, 1}};
// ((2, 16) and (16, 2) would also hold 128 matrix registers but spill at the 256-VGPR budget of 2 waves per SIMD)

}  // namespace

// Picks the decomposition: the narrowest column class that holds n1, then the FEWEST rows per wave that still fits the
// launch on `max_wgs` workgroups (more workgroups = more CUs streaming the distance matrix in and the scores out).
// Returns 0 when the problem does not fit on the chip (the caller falls back to the streaming path).
int ot_resident_plan(int batch, int n0, int n1, int max_wgs, int* nch, int* rpw, int* G, int single) {
    if (batch <= 0 || n0 <= 0 || n1 <= 0) return 0;
    int cls = 0;
    for (const Shape& s : kShapes)
        if (256 * s.nch >= n1) { cls = s.nch; break; }            // kShapes ascends in nch, then in rpw
    if (!cls) return 0;
    if (cls > 8 && !single) return 0;                             // the wide shapes sum their rows in another order: one pair per call only
    for (const Shape& s : kShapes) {
        if (s.nch != cls) continue;
        const int g = (n0 + 8 * s.rpw - 1) / (8 * s.rpw);
        if ((long)g * batch <= max_wgs) { *nch = s.nch; *rpw = s.rpw; *G = g; return 1; }
    }
    return 0;        // the widest row count of the class still needs more workgroups than there are CUs
}

size_t ot_resident_ldx(int nch) { return (size_t)256 * nch + 4; }

bool ot_resident_hier_ok(int nch, int rpw, int G, int batch) { return rpw == 4 && nch >= 5 && nch <= 8 && !(G & 1) && G / 2 <= 32 && batch <= 4; }

hipError_t launch_ot_resident(const OtResidentParams& p, int nch, int rpw, hipStream_t stream) {
#define IMP_OTR(N, R) if (nch == N && rpw == R) return launch_one<N, R>(p, stream)
    IMP_OTR(2, 4); IMP_OTR(2, 8); IMP_OTR(3, 4); IMP_OTR(3, 8); IMP_OTR(4, 4); IMP_OTR(4, 8);
    IMP_OTR(5, 4); IMP_OTR(6, 4); IMP_OTR(8, 4); IMP_OTR(12, 2); IMP_OTR(16, 1);
#undef IMP_OTR
    return hipErrorInvalidValue;
}
