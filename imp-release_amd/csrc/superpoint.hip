// SuperPoint front-end for gfx950: image -> keypoints, scores, descriptors, all on the GPU.
// Reference: nets/superpoint.py:49-63 (simple_nms), :66-79 (remove_borders, top_k_keypoints), :82-94 (sample_descriptors),
// :97-137 (layers), :170-232 (forward), :140-168 (extract).
//
// Data layout: activations are NHWC float32 in HBM ([B][H][W][C], channels contiguous), so a 3x3 convolution is an implicit
// GEMM whose K runs over (tap, channel) with 16-channel MFMA k-steps reading contiguous channels of a shifted pixel:
//
//   sp_conv1a_kernel     1 -> 64 channels, 9 taps on the VALU (K = 9 is no matrix shape), fused bias + ReLU
//   sp_convp_kernel      every other convolution: split-half f16x3 MFMA (v_mfma_f32_32x32x16_f16, x = hi + lo, products lo.hi + hi.lo + hi.hi, fp32
//                        accumulate - the arithmetic of gemm_f32.hip, fp32-level results) as a persistent producer / consumer kernel (see there); the first layer
//                        computes conv1a on the way; epilogue: bias, ReLU and the 2x2 max-pool.  (Rounds 3-4 also kept the one-tile-per-workgroup kernel, a
//                        VALU detector head and the unfused first layer behind environment switches: removed in round 6, profiles/r04.)
//   sp_softmax_shuffle   softmax over the 65 bins of the convPb logits, dustbin dropped, 8x8 pixel shuffle
//   sp_nms_kernel        the whole simple_nms chain (5 max-pools of radius R) for a 32 x 32 tile out of one LDS image with a 5R halo
//   sp_rowcount / sp_scan / sp_compact   ordered compaction = torch.nonzero order (row major) with the border filter folded in
//   sp_topk_select / sp_topk_rank   top_k_keypoints: radix select of the k-th score on one CU, rank sort of the survivors on many
//   sp_sample_kernel     sample_descriptors: one wave per keypoint, bilinear taps of the L2-normalised dense map, re-normalised
//
// LDS bank check for the A-fragment reads (cdna_hip_programming.md section 2: ds_read_b128 is served in the lane groups
// {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32)): lane l of a fragment reads pixel (x = 2 (l >> 2) + (l & 1), y = (l >> 1) & 1);
// with 9 sixteen-byte slots per pixel and a row pitch = 8 slots (mod 16) each group's 16 addresses fall in 16 distinct slots.
#include "../../include/imp_hip.h"
#include "imp_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int imp_fail(int code, const char* msg);   // context.hip: sets imp_last_error()

namespace {

constexpr int TH = 8, TW = 16;             // output pixels of one workgroup tile (before pooling)
constexpr int PSLOT = 9;                   // 16-byte slots per pixel in an LDS plane: 64 halves + 8 halves of padding

template <int TAPS, int TWc>
struct GeoT {
    static constexpr int HALO = TAPS == 9 ? 1 : 0;
    static constexpr int LH = TH + 2 * HALO, LW = TWc + 2 * HALO;
    static constexpr int ROW_SLOTS = ((LW * PSLOT + 7) / 16) * 16 + 8;   // >= LW * PSLOT and = 8 (mod 16)
    static constexpr int PLANE = LH * ROW_SLOTS * 16;                    // bytes of one half plane
    static constexpr int LDS = 2 * PLANE;
};
template <int TAPS>
using Geo = GeoT<TAPS, TW>;

struct SpConvParams {
    const float* in;      // NHWC [B][H][W][in_ld], channels in_c0 .. in_c0 + cin
    int in_ld, in_c0, cin;
    int B, H, W;          // spatial size of the input = of the convolution output before pooling
    const u32x4* wf;      // [cout / 32][ksteps][hi | lo][64 lanes] fragments, kstep = (chunk * TAPS + tap) * 4 + cc
    const float* bias;
    int cout;             // multiple of 64
    float* out;           // NHWC [B][Ho][Wo][out_ld]
    int out_ld;
    int relu;
    int tiles_x, tiles_y;
    const float* img;     // FIRST: the image [B][H][W]; the input tile is relu(conv1a(img)) computed while staging
    const float* w1a;     // [9][64]
    const float* b1a;
    unsigned long long* prof;   // probe (IMP_SP_PROF): [block][4] cycle stamps of wave 0: start, staged, K loop done, stored
};

// bijective XCD-aware remap of a linear block id (hardware places block b on XCD b % 8): each XCD gets a contiguous chunk of
// the logical order, in which the channel tiles of one pixel tile and neighbouring pixel tiles (shared halo) are adjacent
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = lin % 8, idx = lin / 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}



// ---------------------------------------------------------------------------------------------------------------------------
// Persistent producer / consumer form of the convolution (the default): one workgroup of 8 waves per CU loops over its share
// of the (tile, 64-channel group) jobs.  Waves 4-7 are LOADERS - they stage the input tile of unit u + 1 (unit = one 64-channel
// input chunk of a job) into the second LDS buffer, for the first layer computing conv1a on the way - while waves 0-3 run the
// MFMA K loop of unit u out of the first; one barrier per unit swaps the roles of the buffers.  The two wave kinds share each
// SIMD, so the loaders' global loads, conversions and LDS writes fill the issue slots the MFMA chain leaves free (measured on
// the one-tile-per-workgroup kernel above: staging + epilogue were 45-55 % of a workgroup's life and idle time for the matrix
// pipe).  The MFMA operands are swapped with respect to that kernel - weights first - so an accumulator register holds four
// consecutive CHANNELS of the lane's pixel: the epilogue writes 16-byte vectors (8 stores per thread instead of 32) and the 2x2
// max-pool is two DPP quad permutes (the four pixels of a pooling window are four adjacent lanes).
__device__ __forceinline__ float sp_quad_max(float v) {
    int x = __builtin_bit_cast(int, v);
    float a = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v = fmaxf(v, a);
    x = __builtin_bit_cast(int, v);
    a = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true));            // quad_perm [2,3,0,1]
    return fmaxf(v, a);
}

template <int TAPS, int POOL, int FIRST, int TWc>
__global__ __launch_bounds__(512) void sp_convp_kernel(const SpConvParams p, int njobs) {
    using G = GeoT<TAPS, TWc>;
    constexpr int MI = TWc / 8;                  // 32-row MFMA fragments per wave: tile = 8 x TWc pixels = 2 waves x MI x 32
    constexpr int QW = TWc / 2;                  // pooling windows per tile row
    constexpr int PATCH = FIRST ? 1024 : 0;
    constexpr int BUF = G::LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];       // [2][BUF] tiles, then [2][PATCH] image patches
    const int tid = threadIdx.x;
    const bool loader = tid >= 256;
    const int lt = tid & 255, lane = tid & 63, wave = lt >> 6;
    const int H = p.H, W = p.W;
    const int nchunk = p.cin >> 6, nt = p.cout >> 6;
    const int nwg = gridDim.x;
    const int myjobs = (njobs - (int)blockIdx.x + nwg - 1) / nwg;
    const int U = myjobs * nchunk;

    auto decode = [&](int unit, int& b, int& y0, int& x0, int& ntile, int& chunk) {
        const int job = unit / nchunk;
        chunk = unit - job * nchunk;
        int z = xcd_remap((int)blockIdx.x + job * nwg, njobs);     // (nwg is a multiple of 8 whenever a workgroup has a second job)
        ntile = z % nt; z /= nt;
        const int tx = z % p.tiles_x; z /= p.tiles_x;
        const int ty = z % p.tiles_y;
        b = z / p.tiles_y;
        y0 = ty * TH; x0 = tx * TWc;
    };

    // ------------------------------------------------------------------------------------------------ loader side
    constexpr int NPIX = G::LH * G::LW;
    constexpr int NPASS = (NPIX + 15) / 16;
    const int sp_pix = lt >> 4, sp_c = (lt & 15) * 4;
    constexpr int PW = G::LW + 2, PH = G::LH + 2;
    auto load_patch = [&](int unit) {                 // FIRST: the image patch under the tile of `unit`, zero outside the image
        int b, y0, x0, ntile, chunk;
        decode(unit, b, y0, x0, ntile, chunk);
        float* patch = reinterpret_cast<float*>(sp_smem + 2 * BUF + (unit & 1) * PATCH);
        if (lt < PW * PH) {
            const int py = lt / PW, px = lt - py * PW;
            const int yy = y0 - G::HALO - 1 + py, xx = x0 - G::HALO - 1 + px;
            patch[lt] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? p.img[((size_t)b * H + yy) * W + xx] : 0.f;
        }
    };
    auto stage = [&](int unit) {
        int b, y0, x0, ntile, chunk;
        decode(unit, b, y0, x0, ntile, chunk);
        unsigned char* buf = sp_smem + (unit & 1) * BUF;
        if (FIRST) {
            f32x4 wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(p.w1a + t * 64 + sp_c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b1a + sp_c);
            const float* patch = reinterpret_cast<const float*>(sp_smem + 2 * BUF + (unit & 1) * PATCH);
#pragma unroll 2
            for (int s = 0; s < NPASS; ++s) {
                const int pix = s * 16 + sp_pix;
                if (pix >= NPIX) break;
                const int ly = pix / G::LW, lx = pix - ly * G::LW;
                const int gy = y0 - G::HALO + ly, gx = x0 - G::HALO + lx;
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    a = bb;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float v = patch[(ly + t / 3) * PW + lx + t % 3];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = fmaf(v, wv[t][e], a[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f);
                }
                u32x2 hi, lo;
                unsigned ua, uc;
                imp_split2(a[0], a[1], ua, uc); hi[0] = ua; lo[0] = uc;
                imp_split2(a[2], a[3], ua, uc); hi[1] = ua; lo[1] = uc;
                unsigned char* dst = buf + (ly * G::ROW_SLOTS + lx * PSLOT) * 16 + sp_c * 2;
                *reinterpret_cast<u32x2*>(dst) = hi;
                *reinterpret_cast<u32x2*>(dst + G::PLANE) = lo;
            }
        } else {
            const float* src = p.in + p.in_c0 + chunk * 64 + sp_c;
            f32x4 v[NPASS];
#pragma unroll
            for (int s = 0; s < NPASS; ++s) {
                const int pix = s * 16 + sp_pix;
                const int ly = pix / G::LW, lx = pix - ly * G::LW;
                const int gy = y0 - G::HALO + ly, gx = x0 - G::HALO + lx;
                const bool ok = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                v[s] = f32x4{0.f, 0.f, 0.f, 0.f};                                   // zero padding of the convolution
                if (ok) v[s] = *reinterpret_cast<const f32x4*>(src + ((size_t)(b * H + gy) * W + gx) * p.in_ld);
            }
#pragma unroll
            for (int s = 0; s < NPASS; ++s) {
                const int pix = s * 16 + sp_pix;
                if (pix < NPIX) {
                    const int ly = pix / G::LW, lx = pix - ly * G::LW;
                    u32x2 hi, lo;
                    unsigned a, c;
                    imp_split2(v[s][0], v[s][1], a, c); hi[0] = a; lo[0] = c;
                    imp_split2(v[s][2], v[s][3], a, c); hi[1] = a; lo[1] = c;
                    unsigned char* dst = buf + (ly * G::ROW_SLOTS + lx * PSLOT) * 16 + sp_c * 2;
                    *reinterpret_cast<u32x2*>(dst) = hi;
                    *reinterpret_cast<u32x2*>(dst + G::PLANE) = lo;
                }
            }
        }
    };

    // ------------------------------------------------------------------------------------------------ consumer side
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;
    int aoff[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = 32 * MI * wm + 32 * i + (lane & 31);
        const int q = m >> 2, r = m & 3;
        const int py = 2 * (q / QW) + (r >> 1), px = 2 * (q % QW) + (r & 1);
        aoff[i] = (py * G::ROW_SLOTS + px * PSLOT) * 16 + half * 16;
    }
    const int ksteps = nchunk * TAPS * 4;
    auto wbase = [&](int unit) -> const u32x4* {
        int b, y0, x0, ntile, chunk;
        decode(unit, b, y0, x0, ntile, chunk);
        return p.wf + ((size_t)(ntile * 2 + wn) * ksteps + chunk * TAPS * 4) * 128 + lane;
    };
    f32x16 acc[MI];
    u32x4 bh[4], bl[4];
    unsigned long long pa = 0, pb = 0, pc = 0, t0 = 0, t1 = 0;      // probe: loader: staging / barrier wait; consumer: K loop / epilogue / wait
    const bool prof = p.prof != nullptr;

    // ------------------------------------------------------------------------------------------------ pipeline
    if (FIRST) {
        if (loader) load_patch(0);
        __syncthreads();
    }
    if (loader) {
        stage(0);
        if (FIRST && 1 < U) load_patch(1);
    } else {
        const u32x4* w0 = wbase(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bh[c] = w0[c * 128]; bl[c] = w0[c * 128 + 64]; }
    }
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < U; ++u) {
        if (prof) t0 = __builtin_readcyclecounter();
        if (loader) {
            if (u + 1 < U) stage(u + 1);
            if (FIRST && u + 2 < U) load_patch(u + 2);
            if (prof) { t1 = __builtin_readcyclecounter(); pa += t1 - t0; }
        } else {
            int b, y0, x0, ntile, chunk;
            decode(u, b, y0, x0, ntile, chunk);
            const unsigned char* buf = sp_smem + (u & 1) * BUF;
            if (chunk == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            }
            const u32x4* wp = wbase(u);
            const u32x4* wfollow = u + 1 < U ? wbase(u + 1) : wp;        // first tap of the next unit (or a harmless re-load)
            constexpr int NS = TAPS * 4;
            f16x8 fah[2][MI], fal[2][MI];
            auto load_frag = [&](int st, int fb) {
                const int tap = st >> 2, c = st & 3;
                const int toff = TAPS == 9 ? ((tap / 3) * G::ROW_SLOTS + (tap % 3) * PSLOT) * 16 : 0;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    fah[fb][i] = *reinterpret_cast<const f16x8*>(buf + aoff[i] + toff + c * 32);
                    fal[fb][i] = *reinterpret_cast<const f16x8*>(buf + G::PLANE + aoff[i] + toff + c * 32);
                }
            };
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
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fal[st & 1][i], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fah[st & 1][i], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fah[st & 1][i], acc[i], 0, 0, 0);
                if (c == 3) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { bh[cc] = nh[cc]; bl[cc] = nl[cc]; }
                }
            }
            if (prof) { t1 = __builtin_readcyclecounter(); pa += t1 - t0; t0 = t1; }
            if (chunk == nchunk - 1) {
                // epilogue: register r of fragment i = channel cb + 4 half + (r & 3) + 8 (r >> 2) of the lane's pixel m = 64 wm + 32 i + lane % 32
                const int cb = ntile * 64 + wn * 32 + 4 * half;
                f32x4 bv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + cb + 8 * g);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int m = 32 * MI * wm + 32 * i + (lane & 31);
                    const int q = m >> 2, r = m & 3;
                    if (POOL) {
                        const int Ho = H / 2, Wo = W / 2;
                        f32x4 sel = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = sp_quad_max(acc[i][4 * g + e]) + bv[g][e];
                                if (p.relu) v[e] = fmaxf(v[e], 0.f);
                            }
                            if (r == g) sel = v;                               // lane r of the quad stores channel group g = r
                        }
                        const int py = y0 / 2 + q / QW, px = x0 / 2 + q % QW;
                        if (py < Ho && px < Wo)
                            *reinterpret_cast<f32x4*>(p.out + ((size_t)(b * Ho + py) * Wo + px) * p.out_ld + cb + 8 * r) = sel;
                    } else {
                        const int y = y0 + 2 * (q / QW) + (r >> 1), x = x0 + 2 * (q % QW) + (r & 1);
                        if (y < H && x < W) {
                            float* o = p.out + ((size_t)(b * H + y) * W + x) * p.out_ld + cb;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] = acc[i][4 * g + e] + bv[g][e];
                                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                                }
                                *reinterpret_cast<f32x4*>(o + 8 * g) = v;
                            }
                        }
                    }
                }
            }
            if (prof) { t1 = __builtin_readcyclecounter(); pb += t1 - t0; }
        }
        __syncthreads();
        if (prof) pc += __builtin_readcyclecounter() - t1;
    }
    if (prof && (tid == 0 || tid == 256)) {
        unsigned long long* o = p.prof + (size_t)blockIdx.x * 8 + (loader ? 4 : 0);
        o[0] = pa; o[1] = pb; o[2] = pc; o[3] = (unsigned long long)U;
    }
}

// conv1a (nets/superpoint.py:120,142): one input channel, 64 output channels; 16 threads per pixel, 4 channels each
__global__ __launch_bounds__(256) void sp_conv1a_kernel(const float* __restrict__ img, const float* __restrict__ w /*[9][64]*/,
                                                        const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t pix = gid >> 4;
    const int c = (int)(gid & 15) * 4;
    if (pix >= (size_t)B * H * W) return;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const float* im = img + (pix - (size_t)y * W - x);       // start of this image
    f32x4 a = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? im[(size_t)yy * W + xx] : 0.f;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + t * 64 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = fmaf(v, wv[e], a[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f);
    *reinterpret_cast<f32x4*>(out + pix * 64 + c) = a;
}

// detector head tail (nets/superpoint.py:193-198): convPb (1x1, 256 -> 65) + softmax(65) + drop the dustbin + 8x8 pixel shuffle
__global__ __launch_bounds__(256) void sp_softmax_shuffle_kernel(const float* __restrict__ logits /*[npix][128]*/, float* __restrict__ scores,
                                                                 int npix, int h, int w) {
    const int lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= npix) return;
    const float l = logits[(size_t)pi * 128 + lane];
    const float ld = logits[(size_t)pi * 128 + 64];
    float mx = fmaxf(l, ld);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = expf(l - mx), ed = expf(ld - mx);
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    sum += ed;
    const int xx = pi % w, yy = (pi / w) % h, b = pi / (w * h);
    scores[((size_t)b * h * 8 + yy * 8 + (lane >> 3)) * (size_t)(w * 8) + xx * 8 + (lane & 7)] = e / sum;
}

// simple_nms (nets/superpoint.py:49-63) for one 32 x 32 tile: every max-pool of the chain is evaluated on the whole LDS image
// (tile + 5R halo, -inf outside the image = max_pool2d's padding); values within k R of the image edge are wrong after k pools,
// the tile itself depends on 5 pools.
// tile edge: 32 (16 above radius 6, where 32 + 10 R squared no longer fits the LDS)
__device__ __forceinline__ void sp_pool(const float* in, float* tmp, float* out, int E, int R, int tid) {
    const int n = E * E;
    for (int i = tid; i < n; i += 256) {
        const int y = i / E, x = i - y * E;
        float m = -INFINITY;
        const int lo = max(x - R, 0), hi = min(x + R, E - 1);
        for (int xx = lo; xx <= hi; ++xx) m = fmaxf(m, in[y * E + xx]);
        tmp[i] = m;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int y = i / E, x = i - y * E;
        float m = -INFINITY;
        const int lo = max(y - R, 0), hi = min(y + R, E - 1);
        for (int yy = lo; yy <= hi; ++yy) m = fmaxf(m, tmp[yy * E + x]);
        out[i] = m;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void sp_nms_kernel(const float* __restrict__ sin, float* __restrict__ sout, int Hs, int Ws, int R, int NT, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int E = NT + 10 * R, n = E * E;
    float* s = reinterpret_cast<float*>(sp_smem);
    float* a = s + n;
    float* t = a + n;
    float* P = t + n;
    unsigned char* msk = reinterpret_cast<unsigned char*>(P + n);
    const int tid = threadIdx.x;
    int z = blockIdx.x;
    const int tx = z % tiles_x; z /= tiles_x;
    const int ty = z % tiles_y;
    const int b = z / tiles_y;
    const int y0 = ty * NT - 5 * R, x0 = tx * NT - 5 * R;
    const float* src = sin + (size_t)b * Hs * Ws;
    for (int i = tid; i < n; i += 256) {
        const int y = i / E, x = i - y * E;
        const int gy = y0 + y, gx = x0 + x;
        s[i] = (gy >= 0 && gy < Hs && gx >= 0 && gx < Ws) ? src[(size_t)gy * Ws + gx] : -INFINITY;
    }
    __syncthreads();
    sp_pool(s, t, P, E, R, tid);
    for (int i = tid; i < n; i += 256) msk[i] = (s[i] > -INFINITY && s[i] == P[i]) ? 1 : 0;
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
        for (int i = tid; i < n; i += 256) a[i] = msk[i] ? 1.f : 0.f;
        __syncthreads();
        sp_pool(a, t, P, E, R, tid);
        for (int i = tid; i < n; i += 256) {
            const bool supp = P[i] > 0.f;
            a[i] = supp ? (s[i] > -INFINITY ? 0.f : -INFINITY) : s[i];
            msk[i] = (unsigned char)(msk[i] | (supp ? 2 : 0));           // bit 1: suppressed in this round
        }
        __syncthreads();
        sp_pool(a, t, P, E, R, tid);
        for (int i = tid; i < n; i += 256) {
            const bool supp = (msk[i] & 2) != 0;
            const bool nw = s[i] > -INFINITY && a[i] == P[i];
            msk[i] = (unsigned char)((msk[i] & 1) | ((nw && !supp) ? 1 : 0));
        }
        __syncthreads();
    }
    float* dst = sout + (size_t)b * Hs * Ws;
    for (int i = tid; i < NT * NT; i += 256) {
        const int y = i / NT, x = i - y * NT;
        const int gy = ty * NT + y, gx = tx * NT + x;
        if (gy < Hs && gx < Ws) {
            const int j = (y + 5 * R) * E + x + 5 * R;
            dst[(size_t)gy * Ws + gx] = msk[j] ? s[j] : 0.f;
        }
    }
}


__device__ __forceinline__ bool sp_keep(float v, int x, int y, int Hs, int Ws, float thr, int border) {
    return v > thr && y >= border && y < Hs - border && x >= border && x < Ws - border;
}

// Fast path of simple_nms for radius 1..6 (compile-time R): same chain, same LDS image, but every pool is a register-blocked
// separable pass - a thread produces 8 consecutive outputs from 8 + 2R inputs it holds in registers (horizontal: aligned float4
// reads of a row padded with -inf columns; vertical: a column of the intermediate image, which has R rows of -inf above and
// below) - and the vertical pass hands each pooled value straight to the comparison that consumes it, so the pooled image is
// never stored.  3 float images + 1 byte mask = 74 KB for R = 4 (2 workgroups per CU).
template <int R>
struct NmsGeo {
    static constexpr int T = 32;
    static constexpr int E = ((T + 10 * R + 7) / 8) * 8;     // LDS image edge (tile + 5R halo, rounded up to the 8-wide blocks)
    static constexpr int RP = ((R + 3) / 4) * 4;             // -inf columns on both sides of s / a rows (16-byte aligned)
    static constexpr int ES = E + 2 * RP;
    static constexpr int TR = E + 2 * R;                     // rows of the intermediate image
    static constexpr int LDS = (2 * E * ES + TR * E) * 4 + E * E;
};

template <int R, class F>
__device__ __forceinline__ void nms_pool(const float* in, float* t, int tid, F&& f) {
    using G = NmsGeo<R>;
    constexpr int E = G::E, RP = G::RP, ES = G::ES, SEG = E / 8;
    for (int seg = tid; seg < E * SEG; seg += 256) {
        const int y = seg / SEG, x0 = (seg - y * SEG) * 8;
        float v[8 + 2 * RP];
        const float* row = in + y * ES + x0;                 // padded index x0 = logical x0 - RP
#pragma unroll
        for (int j = 0; j < (8 + 2 * RP) / 4; ++j) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(row + 4 * j);
            v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3];
        }
        f32x4 o[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float m = v[RP - R + i];
#pragma unroll
            for (int j = 1; j <= 2 * R; ++j) m = fmaxf(m, v[RP - R + i + j]);
            o[i >> 2][i & 3] = m;
        }
        float* dst = t + (y + R) * E + x0;
        *reinterpret_cast<f32x4*>(dst) = o[0];
        *reinterpret_cast<f32x4*>(dst + 4) = o[1];
    }
    __syncthreads();
    for (int seg = tid; seg < E * SEG; seg += 256) {
        const int yb = seg / E, x = seg - yb * E;
        float v[8 + 2 * R];
#pragma unroll
        for (int j = 0; j < 8 + 2 * R; ++j) v[j] = t[(yb * 8 + j) * E + x];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float m = v[i];
#pragma unroll
            for (int j = 1; j <= 2 * R; ++j) m = fmaxf(m, v[i + j]);
            f(yb * 8 + i, x, m);
        }
    }
    __syncthreads();
}

template <int R>
__global__ __launch_bounds__(256) void sp_nms_fast_kernel(const float* __restrict__ sin, float* __restrict__ sout, int Hs, int Ws, int tiles_x, int tiles_y,
                                                          float thr, int border, int* __restrict__ tilecount /*[B][Hs][tiles_x]*/) {
    using G = NmsGeo<R>;
    constexpr int E = G::E, RP = G::RP, ES = G::ES, TR = G::TR, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    float* s = reinterpret_cast<float*>(sp_smem);            // [E][ES]
    float* a = s + E * ES;                                   // [E][ES]
    float* t = a + E * ES;                                   // [TR][E]
    unsigned char* msk = reinterpret_cast<unsigned char*>(t + TR * E);
    const int tid = threadIdx.x;
    int z = blockIdx.x;
    const int tx = z % tiles_x; z /= tiles_x;
    const int ty = z % tiles_y;
    const int b = z / tiles_y;
    const int y0 = ty * T - 5 * R, x0 = tx * T - 5 * R;
    const float* src = sin + (size_t)b * Hs * Ws;
    for (int i = tid; i < E * ES; i += 256) {
        const int y = i / ES, xp = i - y * ES;
        const int x = xp - RP;
        const int gy = y0 + y, gx = x0 + x;
        const bool in = x >= 0 && x < E && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
        s[i] = in ? src[(size_t)gy * Ws + gx] : -INFINITY;
        a[i] = -INFINITY;                                    // the column pads stay -inf; the interior is rewritten per pool
    }
    for (int i = tid; i < R * E; i += 256) {
        t[i] = -INFINITY;
        t[(E + R) * E + i] = -INFINITY;
    }
    __syncthreads();
    nms_pool<R>(s, t, tid, [&](int y, int x, float P) {
        const float v = s[y * ES + RP + x];
        const bool m = v > -INFINITY && v == P;
        msk[y * E + x] = m ? 1 : 0;
        a[y * ES + RP + x] = m ? 1.f : 0.f;
    });
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        nms_pool<R>(a, t, tid, [&](int y, int x, float P) {
            const float v = s[y * ES + RP + x];
            const bool supp = P > 0.f;
            a[y * ES + RP + x] = supp ? (v > -INFINITY ? 0.f : -INFINITY) : v;
            if (supp) msk[y * E + x] |= 2;
        });
        nms_pool<R>(a, t, tid, [&](int y, int x, float P) {
            const float v = s[y * ES + RP + x];
            const unsigned char mk = msk[y * E + x];
            const bool nw = v > -INFINITY && a[y * ES + RP + x] == P;
            const bool m = (mk & 1) || (nw && !(mk & 2));
            msk[y * E + x] = m ? 1 : 0;
            a[y * ES + RP + x] = m ? 1.f : 0.f;
        });
    }
    float* dst = sout + (size_t)b * Hs * Ws;
    for (int i = tid; i < T * T; i += 256) {
        const int y = i / T, x = i - y * T;
        const int gy = ty * T + y, gx = tx * T + x;
        float v = 0.f;
        if (gy < Hs && gx < Ws) {
            const int ly = y + 5 * R, lx = x + 5 * R;
            v = msk[ly * E + lx] ? s[ly * ES + RP + lx] : 0.f;
            dst[(size_t)gy * Ws + gx] = v;
        }
        // the keypoint candidates of this row segment (T = 32 = half a wave): what sp_rowcount_kernel would count
        const bool keep = gy < Hs && gx < Ws && sp_keep(v, gx, gy, Hs, Ws, thr, border);
        const unsigned long long bal = __ballot(keep);
        if ((tid & 31) == 0 && gy < Hs) tilecount[((size_t)b * Hs + gy) * tiles_x + tx] = __popcll((tid & 32) ? (bal >> 32) : (bal & 0xFFFFFFFFull));
    }
}

template <int R>
int launch_nms_fast(const float* in, float* out, int B, int Hs, int Ws, float thr, int border, int* tilecount, hipStream_t st);

// keypoint extraction (nets/superpoint.py:203-211): torch.nonzero(s > threshold) order = row major, border filter of :66-71

__global__ __launch_bounds__(64) void sp_rowcount_kernel(const float* __restrict__ nms, int Hs, int Ws, float thr, int border, int* __restrict__ rowcount) {
    const int y = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float* row = nms + ((size_t)b * Hs + y) * Ws;
    int cnt = 0;
    for (int x0 = 0; x0 < Ws; x0 += 64) {
        const int x = x0 + lane;
        const bool k = x < Ws && sp_keep(row[x], x, y, Hs, Ws, thr, border);
        cnt += __popcll(__ballot(k));
    }
    if (lane == 0) rowcount[b * Hs + y] = cnt;
}

__global__ __launch_bounds__(1024) void sp_scan_kernel(const int* __restrict__ rowcount, const int* __restrict__ tilecount, int tiles_x, int Hs,
                                                       int* __restrict__ rowoff, int* __restrict__ count) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < Hs; base += 1024) {
        const int i = base + tid;
        int v = 0;
        if (i < Hs) {
            if (tilecount) {                                   // per-tile counts of the fast NMS kernel
                for (int t = 0; t < tiles_x; ++t) v += tilecount[((size_t)b * Hs + i) * tiles_x + t];
            } else {
                v = rowcount[b * Hs + i];
            }
        }
        part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int add = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < Hs) rowoff[b * Hs + i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) count[b] = carry;
}

__global__ __launch_bounds__(64) void sp_compact_kernel(const float* __restrict__ nms, int Hs, int Ws, float thr, int border,
                                                        const int* __restrict__ rowoff, float* __restrict__ kp, float* __restrict__ sc, size_t cap) {
    const int y = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float* row = nms + ((size_t)b * Hs + y) * Ws;
    int off = rowoff[b * Hs + y];
    float* kpb = kp + (size_t)b * cap * 2;
    float* scb = sc + (size_t)b * cap;
    for (int x0 = 0; x0 < Ws; x0 += 64) {
        const int x = x0 + lane;
        const float v = x < Ws ? row[x] : 0.f;
        const bool k = x < Ws && sp_keep(v, x, y, Hs, Ws, thr, border);
        const unsigned long long bal = __ballot(k);
        if (k) {
            const int r = off + __popcll(bal & ((1ull << lane) - 1ull));
            kpb[2 * (size_t)r] = (float)x;                              // (x, y): torch.flip of (row, col), nets/superpoint.py:220
            kpb[2 * (size_t)r + 1] = (float)y;
            scb[r] = v;
        }
        off += __popcll(bal);
    }
}

// top_k_keypoints (nets/superpoint.py:74-79): k >= n keeps everything in nonzero order; otherwise the k best scores, descending
// (torch.topk), equal scores in index order.  Keys = score bits (scores are positive floats: the bit pattern is monotone) : ~index.
//   sp_topk_select_kernel (one workgroup of 1024 threads per image): n <= k copies; n <= 2048 hands every key on; otherwise a
//     radix select of the k-th largest score (digits of 12 / 10 / 10 bits from the top, LDS histogram, parallel suffix scan)
//     cuts the candidates to exactly k keys.
//   sp_topk_rank_kernel (16 keys per workgroup, 16 threads per key, so the chip - not one CU - does the m^2 comparisons): rank
//     sort, a key's final position is the number of larger keys; no barriers in the loop.
constexpr int TOPK_CAP = 16384;
constexpr int TOPK_DIRECT = 2048;

// inclusive prefix sum over the 1024 threads of the block (wave shuffles + one LDS hop); all threads must call
__device__ __forceinline__ unsigned sp_block_scan(unsigned v, unsigned* wave_tot /*[16]*/, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    __syncthreads();                       // previous users of wave_tot are done
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    unsigned base = 0;
    for (int wv = 0; wv < wave; ++wv) base += wave_tot[wv];
    return x + base;
}

__device__ __forceinline__ unsigned long long sp_key(unsigned score_bits, int i) {
    return ((unsigned long long)score_bits << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
}

__global__ __launch_bounds__(1024) void sp_topk_select_kernel(const float* __restrict__ kp, const float* __restrict__ sc, const int* __restrict__ count,
                                                              size_t cap, int k, float* __restrict__ kp_out, float* __restrict__ sc_out,
                                                              int* __restrict__ count_out, unsigned long long* __restrict__ selkeys /*[B][TOPK_CAP]*/,
                                                              int* __restrict__ sel_m) {
    __shared__ unsigned hist[4096];
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned sel_prefix, sel_remaining, sel_count, eq_taken;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = count[b];
    const float* kpb = kp + (size_t)b * cap * 2;
    const float* scb = sc + (size_t)b * cap;
    unsigned long long* keys = selkeys + (size_t)b * TOPK_CAP;
    if (k < 0 || k >= n) {
        float* kpo = kp_out + (size_t)b * cap * 2;
        float* sco = sc_out + (size_t)b * cap;
        for (int i = tid; i < n; i += 1024) {
            kpo[2 * (size_t)i] = kpb[2 * (size_t)i];
            kpo[2 * (size_t)i + 1] = kpb[2 * (size_t)i + 1];
            sco[i] = scb[i];
        }
        if (tid == 0) { count_out[b] = n; sel_m[b] = 0; }
        return;
    }
    if (n <= TOPK_DIRECT) {
        for (int i = tid; i < n; i += 1024) keys[i] = sp_key(__float_as_uint(scb[i]), i);
        if (tid == 0) { count_out[b] = k; sel_m[b] = n; }
        return;
    }
    if (tid == 0) { sel_prefix = 0; sel_remaining = (unsigned)k; }
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 20 : pass == 1 ? 10 : 0;
        const int bits = pass == 0 ? 12 : 10;
        const int nb = 1 << bits;
        for (int i = tid; i < nb; i += 1024) hist[i] = 0;
        __syncthreads();
        const unsigned prefix = sel_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + bits));
        for (int i = tid; i < n; i += 1024) {
            const unsigned u = __float_as_uint(scb[i]);
            if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        // thread t owns the bins nb-1 - per*t - j (descending), so the prefix over threads is the suffix over bins
        const int per = nb >> 10;                     // 4 or 1
        unsigned mine = 0;
        for (int j = 0; j < per; ++j) mine += hist[nb - 1 - per * tid - j];
        const unsigned incl = sp_block_scan(mine, wave_tot, tid);
        const unsigned rem = sel_remaining;
        __syncthreads();                              // everyone has read sel_remaining
        if (incl - mine < rem && rem <= incl) {       // exactly one thread: the k-th largest falls into its bins
            unsigned r = rem - (incl - mine);
            int d = nb - 1 - per * tid;
            for (int j = 0; j < per; ++j, --d) {
                if (hist[d] >= r) break;
                r -= hist[d];
            }
            sel_prefix = prefix | ((unsigned)d << shift);
            sel_remaining = r;                        // how many of the elements with this prefix are still needed
        }
        __syncthreads();
    }
    const unsigned T = sel_prefix;
    const unsigned need_eq = sel_remaining;           // elements == T to take
    if (tid == 0) { sel_count = 0; eq_taken = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {             // elements > T; count the == T ones
        const unsigned u = __float_as_uint(scb[i]);
        if (u > T) {
            const unsigned slot = atomicAdd(&sel_count, 1u);
            if (slot < (unsigned)TOPK_CAP) keys[slot] = sp_key(u, i);
        } else if (u == T) {
            atomicAdd(&eq_taken, 1u);
        }
    }
    __syncthreads();
    const unsigned n_eq = eq_taken, n_gt = sel_count;
    __syncthreads();
    if (n_eq == need_eq) {                            // (the usual case) every element == T is needed: order does not matter
        for (int i = tid; i < n; i += 1024) {
            const unsigned u = __float_as_uint(scb[i]);
            if (u == T) {
                const unsigned slot = atomicAdd(&sel_count, 1u);
                if (slot < (unsigned)TOPK_CAP) keys[slot] = sp_key(u, i);
            }
        }
    } else {                                          // exact ties at the cut: the first need_eq in index order (deterministic)
        unsigned taken = 0;
        for (int base = 0; base < n; base += 1024) {
            const int i = base + tid;
            const bool eq = i < n && __float_as_uint(scb[i]) == T;
            const unsigned incl = sp_block_scan(eq ? 1u : 0u, wave_tot, tid);
            const unsigned rank = taken + incl - 1u;
            if (eq && rank < need_eq && n_gt + rank < (unsigned)TOPK_CAP) keys[n_gt + rank] = sp_key(T, i);
            __syncthreads();
            if (tid == 1023) eq_taken = incl;         // block total of this chunk
            __syncthreads();
            taken += eq_taken;
        }
    }
    if (tid == 0) { count_out[b] = k; sel_m[b] = k; } // n_gt + need_eq == k (the host rejects k > TOPK_CAP)
}

__global__ __launch_bounds__(256) void sp_topk_rank_kernel(const float* __restrict__ kp, size_t cap, int k, const unsigned long long* __restrict__ selkeys,
                                                           const int* __restrict__ sel_m, float* __restrict__ kp_out, float* __restrict__ sc_out) {
    // 16 keys per workgroup, 16 threads per key: thread (key, part) counts the larger keys among every 16th group of 16 keys
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(sp_smem);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int m = sel_m[b];
    if (blockIdx.x * 16 >= m) return;                 // (m == 0: nothing to sort for this image)
    const unsigned long long* src = selkeys + (size_t)b * TOPK_CAP;
    const int m2 = (m + 255) & ~255;
    for (int i = tid; i < m2; i += 256) keys[i] = i < m ? src[i] : 0ull;      // (padding keys are smaller than every real key)
    __syncthreads();
    const int mine_i = blockIdx.x * 16 + (tid >> 4), part = tid & 15;
    const unsigned long long mk = mine_i < m ? keys[mine_i] : ~0ull;
    const unsigned mhi = (unsigned)(mk >> 32), mlo = (unsigned)mk;
    unsigned rank = 0;
    for (int i = part * 16; i < m2; i += 256) {
        u32x4 o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = *reinterpret_cast<const u32x4*>(keys + i + 2 * j);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                 // (little endian: [0] = ~index, [1] = score bits)
            rank += (o[j][1] > mhi || (o[j][1] == mhi && o[j][0] > mlo)) ? 1u : 0u;
            rank += (o[j][3] > mhi || (o[j][3] == mhi && o[j][2] > mlo)) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) rank += __shfl_xor(rank, o);
    if (part == 0 && mine_i < m && rank < (unsigned)k) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(mk & 0xFFFFFFFFull);
        const float* kpb = kp + (size_t)b * cap * 2;
        float* kpo = kp_out + (size_t)b * cap * 2;
        kpo[2 * (size_t)rank] = kpb[2 * (size_t)idx];
        kpo[2 * (size_t)rank + 1] = kpb[2 * (size_t)idx + 1];
        sc_out[(size_t)b * cap + rank] = __uint_as_float((unsigned)(mk >> 32));
    }
}

// sample_descriptors (nets/superpoint.py:82-94): one wave per keypoint, lane = 4 channels (D <= 256, D % 4 == 0).  dmap is the raw
// convDb output NHWC [h][w][D]; F.normalize(p=2, dim=1) of the dense map (:225, eps 1e-12) is applied to the four taps on the fly,
// grid_sample is bilinear with zero padding and the align_corners the caller resolved, the result is normalised again (:92-93).
__device__ __forceinline__ float sp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void sp_sample_kernel(const float* __restrict__ kp, const float* __restrict__ sc, int n, const float* __restrict__ dmap,
                                                        int h, int w, int D, int align_corners, float* __restrict__ kp_out,
                                                        float* __restrict__ sc_out, float* __restrict__ out /*[D][n]*/) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    if (lane < 2) kp_out[2 * (size_t)i + lane] = kp[2 * (size_t)i + lane];
    if (lane == 2) sc_out[i] = sc[i];
    const float s = 8.f;
    float gx = kp[2 * (size_t)i] - s / 2 + 0.5f, gy = kp[2 * (size_t)i + 1] - s / 2 + 0.5f;
    gx = gx / (w * s - s / 2 - 0.5f);
    gy = gy / (h * s - s / 2 - 0.5f);
    gx = gx * 2 - 1;
    gy = gy * 2 - 1;
    float ix, iy;
    if (align_corners) {
        ix = ((gx + 1) / 2) * (w - 1);
        iy = ((gy + 1) / 2) * (h - 1);
    } else {
        ix = ((gx + 1) * w - 1) / 2;
        iy = ((gy + 1) * h - 1) / 2;
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
    const int c = lane * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
        const float wgt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
        if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;             // zero padding (uniform per wave)
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) v = *reinterpret_cast<const f32x4*>(dmap + ((size_t)yy * w + xx) * D + c);
        const float nrm = sqrtf(sp_wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
        const float inv = 1.f / fmaxf(nrm, 1e-12f);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (v[e] * inv) * wgt;
    }
    const float nrm = sqrtf(sp_wave_sum(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3]));
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    if (c < D)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(size_t)(c + e) * n + i] = acc[e] * inv;
}

// dense descriptors of extract() (nets/superpoint.py:163-166): NHWC raw -> NCHW, L2-normalised over channels; one wave per pixel
__global__ __launch_bounds__(256) void sp_dense_desc_kernel(const float* __restrict__ dmap, int npix_per_image, int total, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;
    const int c = lane * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D) v = *reinterpret_cast<const f32x4*>(dmap + (size_t)i * D + c);
    const float nrm = sqrtf(sp_wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    const int b = i / npix_per_image, pi = i - b * npix_per_image;
    if (c < D)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[((size_t)b * D + c + e) * npix_per_image + pi] = v[e] * inv;
}

// ------------------------------------------------------------------------------------------------------------ host side
struct ConvW {
    u32x4* wf = nullptr;
    float* bias = nullptr;
    int cin = 0, cout = 0, taps = 0;
};

struct HostT { std::vector<float> data; };

}  // namespace

struct imp_sp_ctx {
    int device = 0, ddim = 256;
    bool finalized = false;
    std::map<std::string, HostT> raw;
    float* w1a = nullptr;      // [9][64]
    float* b1a = nullptr;
    ConvW c1b, c2a, c2b, c3a, c3b, c4a, c4b, heads, db, pb;
    float* wpb = nullptr;      // [256][65]
    float* bpb = nullptr;
    // workspace of the last detect call
    int B = 0, H = 0, W = 0, h = 0, w = 0;
    size_t capA = 0, capB = 0, cap_small = 0, cap_map = 0, cap_rows = 0, cap_b = 0;
    float *bufA = nullptr, *bufB = nullptr, *dmap = nullptr, *logits = nullptr, *scores = nullptr, *nms = nullptr;
    float *kp0 = nullptr, *sc0 = nullptr, *kp1 = nullptr, *sc1 = nullptr;
    int *rowcount = nullptr, *rowoff = nullptr, *count0 = nullptr, *count1 = nullptr, *sel_m = nullptr, *tilecount = nullptr;
    int* count1_host = nullptr;    // count1 is the device view of this pinned, mapped host array: the kernel's store IS the read-back
    unsigned long long* selkeys = nullptr;
    std::vector<int> counts;
    int align_corners = 1;
    bool detected = false;
};

namespace {

#define SP_TRY(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return imp_fail(IMP_E_HIP, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
    } while (0)

// conv weight [cout][cin][k][k] (+ optional second tensor stacked along cout) -> split-half MFMA B fragments
int pack_conv(imp_sp_ctx* c, const std::vector<std::string>& names, int cin, int taps, ConvW* out) {
    int cout = 0;
    std::vector<const HostT*> ws, bs;
    for (const std::string& nm : names) {
        auto iw = c->raw.find(nm + ".weight"), ib = c->raw.find(nm + ".bias");
        if (iw == c->raw.end() || ib == c->raw.end()) return imp_fail(IMP_E_KEY, ("missing state_dict key: " + nm + ".weight / .bias").c_str());
        const size_t co = ib->second.data.size();
        if (iw->second.data.size() != co * cin * taps) return imp_fail(IMP_E_KEY, ("state_dict tensor " + nm + ".weight has an unexpected size").c_str());
        ws.push_back(&iw->second);
        bs.push_back(&ib->second);
        cout += (int)co;
    }
    if (cout % 64 || cin % 64) return imp_fail(IMP_E_ARG, "convolution channels must be multiples of 64");
    std::vector<float> Wall((size_t)cout * cin * taps), ball(cout);
    {
        size_t o = 0, ob = 0;
        for (size_t i = 0; i < ws.size(); ++i) {
            memcpy(Wall.data() + o, ws[i]->data.data(), ws[i]->data.size() * sizeof(float));
            o += ws[i]->data.size();
            memcpy(ball.data() + ob, bs[i]->data.data(), bs[i]->data.size() * sizeof(float));
            ob += bs[i]->data.size();
        }
    }
    const int nchunk = cin / 64, ksteps = nchunk * taps * 4;
    std::vector<_Float16> frag((size_t)(cout / 32) * ksteps * 2 * 64 * 8);
    for (int nt = 0; nt < cout / 32; ++nt)
        for (int ks = 0; ks < ksteps; ++ks) {
            const int chunk = ks / (taps * 4), tap = (ks / 4) % taps, cc = ks % 4;
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int n = nt * 32 + (lane & 31);
                    const int ch = chunk * 64 + cc * 16 + 8 * (lane >> 5) + e;
                    const float v = Wall[((size_t)n * cin + ch) * taps + tap];
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t base = ((size_t)(nt * ksteps + ks) * 2) * 64 * 8;
                    frag[base + (size_t)lane * 8 + e] = hi;
                    frag[base + 64 * 8 + (size_t)lane * 8 + e] = lo;
                }
        }
    // (a failure below leaves whatever was allocated in *out: imp_sp_finalize / imp_sp_destroy release it through free_conv)
    SP_TRY(hipMalloc(&out->wf, frag.size() * sizeof(_Float16)));
    SP_TRY(hipMemcpy(out->wf, frag.data(), frag.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    SP_TRY(hipMalloc(&out->bias, ball.size() * sizeof(float)));
    SP_TRY(hipMemcpy(out->bias, ball.data(), ball.size() * sizeof(float), hipMemcpyHostToDevice));
    out->cin = cin; out->cout = cout; out->taps = taps;
    return IMP_OK;
}

void free_conv(ConvW& w) {
    if (w.wf) (void)hipFree(w.wf);
    if (w.bias) (void)hipFree(w.bias);
    w = ConvW();
}

template <int TAPS, int POOL, int FIRST, int TWc>
int launch_convp(SpConvParams p, int total, int tiles_x, int ncu, hipStream_t st) {
    p.tiles_x = tiles_x;
    const int nwg = total < ncu ? total : (ncu / 8) * 8;
    constexpr size_t ldsp = 2 * GeoT<TAPS, TWc>::LDS + (FIRST ? 2048 : 0);
    const void* fnp = reinterpret_cast<const void*>(&sp_convp_kernel<TAPS, POOL, FIRST, TWc>);
    SP_TRY(imp_grant_dynamic_lds(fnp, ldsp));
    hipLaunchKernelGGL((sp_convp_kernel<TAPS, POOL, FIRST, TWc>), dim3(nwg), dim3(512), ldsp, st, p, total);
    SP_TRY(hipGetLastError());
    return IMP_OK;
}

template <int TAPS, int POOL, int FIRST>
int launch_conv_t(const SpConvParams& p, hipStream_t st) {
    const int total = p.B * p.tiles_x * p.tiles_y * (p.cout / 64);
    {
        int dev = 0;
        SP_TRY(hipGetDevice(&dev));
        static int ncu_of[64] = {0};                                      // CU count per device, looked up once
        if (dev < 0 || dev >= 64) return imp_fail(IMP_E_ARG, "device index out of range");
        if (!ncu_of[dev]) {
            int n = 0;
            SP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            ncu_of[dev] = n > 0 ? n : 256;
        }
        const int ncu = ncu_of[dev];
        // tile width 16 or 8 pixels: whichever needs less time for the busiest workgroup (a half tile costs ~0.55 of a full one:
        // same loader work per pixel, twice the weight traffic per MFMA) - the layers below 240 x 320 have too few full tiles
        const int tx8 = (p.W + 7) / 8;
        const int jobs16 = total, jobs8 = p.B * tx8 * p.tiles_y * (p.cout / 64);
        const double t16 = (double)((jobs16 + ncu - 1) / ncu), t8 = 0.55 * (double)((jobs8 + ncu - 1) / ncu);
        const bool narrow = t8 < t16;
        return narrow ? launch_convp<TAPS, POOL, FIRST, 8>(p, jobs8, tx8, ncu, st) : launch_convp<TAPS, POOL, FIRST, 16>(p, jobs16, p.tiles_x, ncu, st);
    }
}

int launch_conv(const ConvW& w, const float* in, int in_ld, int in_c0, int B, int H, int W, float* out, int out_ld, int relu, int pool, hipStream_t st) {
    SpConvParams p;
    p.in = in; p.in_ld = in_ld; p.in_c0 = in_c0; p.cin = w.cin;
    p.B = B; p.H = H; p.W = W;
    p.wf = w.wf; p.bias = w.bias; p.cout = w.cout;
    p.out = out; p.out_ld = out_ld; p.relu = relu;
    p.tiles_x = (W + TW - 1) / TW; p.tiles_y = (H + TH - 1) / TH;
    p.img = nullptr; p.w1a = nullptr; p.b1a = nullptr; p.prof = nullptr;
    if (w.taps == 9) return pool ? launch_conv_t<9, 1, 0>(p, st) : launch_conv_t<9, 0, 0>(p, st);
    return launch_conv_t<1, 0, 0>(p, st);
}

// conv1a + conv1b + ReLUs + max-pool in one kernel: the 64-channel full-resolution map never exists in memory
int launch_conv_first(const imp_sp_ctx* c, const float* image, int B, int H, int W, float* out, hipStream_t st);

template <int R>
int launch_nms_fast(const float* in, float* out, int B, int Hs, int Ws, float thr, int border, int* tilecount, hipStream_t st) {
    using G = NmsGeo<R>;
    const int tx = (Ws + G::T - 1) / G::T, ty = (Hs + G::T - 1) / G::T;
    SP_TRY(imp_grant_dynamic_lds(reinterpret_cast<const void*>(&sp_nms_fast_kernel<R>), G::LDS));
    hipLaunchKernelGGL((sp_nms_fast_kernel<R>), dim3(tx * ty * B), dim3(256), G::LDS, st, in, out, Hs, Ws, tx, ty, thr, border, tilecount);
    SP_TRY(hipGetLastError());
    return IMP_OK;
}

int launch_conv_first(const imp_sp_ctx* c, const float* image, int B, int H, int W, float* out, hipStream_t st) {
    const ConvW& w = c->c1b;
    SpConvParams p;
    p.in = nullptr; p.in_ld = 64; p.in_c0 = 0; p.cin = 64;
    p.B = B; p.H = H; p.W = W;
    p.wf = w.wf; p.bias = w.bias; p.cout = w.cout;
    p.out = out; p.out_ld = 64; p.relu = 1;
    p.tiles_x = (W + TW - 1) / TW; p.tiles_y = (H + TH - 1) / TH;
    p.img = image; p.w1a = c->w1a; p.b1a = c->b1a; p.prof = nullptr;
    return launch_conv_t<9, 1, 1>(p, st);
}

void free_ws(imp_sp_ctx* c) {
    for (void* p : {(void*)c->bufA, (void*)c->bufB, (void*)c->dmap, (void*)c->logits, (void*)c->scores, (void*)c->nms, (void*)c->kp0, (void*)c->sc0, (void*)c->kp1,
                    (void*)c->sc1, (void*)c->rowcount, (void*)c->rowoff, (void*)c->count0, (void*)c->sel_m, (void*)c->selkeys, (void*)c->tilecount})
        if (p) (void)hipFree(p);
    if (c->count1_host) (void)hipHostFree(c->count1_host);
    c->count1_host = nullptr; c->tilecount = nullptr;
    c->bufA = c->bufB = c->dmap = c->logits = c->scores = c->nms = c->kp0 = c->sc0 = c->kp1 = c->sc1 = nullptr;
    c->rowcount = c->rowoff = c->count0 = c->count1 = c->sel_m = nullptr;
    c->selkeys = nullptr;
    c->capA = c->capB = c->cap_small = c->cap_map = c->cap_rows = c->cap_b = 0;
}

}  // namespace

extern "C" {

int imp_sp_create(imp_sp_ctx** out, int device, int descriptor_dim) {
    if (!out) return imp_fail(IMP_E_ARG, "imp_sp_create: null output");
    if (descriptor_dim < 64 || descriptor_dim > 256 || descriptor_dim % 64) return imp_fail(IMP_E_ARG, "descriptor_dim must be 64, 128, 192 or 256");
    int ndev = 0;
    SP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return imp_fail(IMP_E_ARG, "imp_sp_create: no such device");
    imp_sp_ctx* c = new imp_sp_ctx();
    c->device = device;
    c->ddim = descriptor_dim;
    *out = c;
    return IMP_OK;
}

void imp_sp_destroy(imp_sp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_ws(c);
    for (ConvW* w : {&c->c1b, &c->c2a, &c->c2b, &c->c3a, &c->c3b, &c->c4a, &c->c4b, &c->heads, &c->db, &c->pb}) free_conv(*w);
    for (void* p : {(void*)c->w1a, (void*)c->b1a, (void*)c->wpb, (void*)c->bpb})
        if (p) (void)hipFree(p);
    delete c;
}

int imp_sp_set_weight(imp_sp_ctx* c, const char* name, const float* data, int64_t count) {
    if (!c || !name || !data || count <= 0) return imp_fail(IMP_E_ARG, "imp_sp_set_weight: bad argument");
    HostT t;
    t.data.assign(data, data + count);
    c->raw[name] = std::move(t);
    c->finalized = false;
    return IMP_OK;
}

int imp_sp_finalize(imp_sp_ctx* c) {
    if (!c) return imp_fail(IMP_E_ARG, "imp_sp_finalize: null context");
    SP_TRY(hipSetDevice(c->device));
    SP_TRY(hipDeviceSynchronize());
    for (ConvW* w : {&c->c1b, &c->c2a, &c->c2b, &c->c3a, &c->c3b, &c->c4a, &c->c4b, &c->heads, &c->db, &c->pb}) free_conv(*w);
    for (float** p : {&c->w1a, &c->b1a, &c->wpb, &c->bpb})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    int rc;
    if ((rc = pack_conv(c, {"conv1b"}, 64, 9, &c->c1b)) || (rc = pack_conv(c, {"conv2a"}, 64, 9, &c->c2a)) ||
        (rc = pack_conv(c, {"conv2b"}, 64, 9, &c->c2b)) || (rc = pack_conv(c, {"conv3a"}, 64, 9, &c->c3a)) ||
        (rc = pack_conv(c, {"conv3b"}, 128, 9, &c->c3b)) || (rc = pack_conv(c, {"conv4a"}, 128, 9, &c->c4a)) ||
        (rc = pack_conv(c, {"conv4b"}, 128, 9, &c->c4b)) || (rc = pack_conv(c, {"convPa", "convDa"}, 128, 9, &c->heads)) ||
        (rc = pack_conv(c, {"convDb"}, 256, 1, &c->db)))
        return rc;
    if (c->c1b.cout != 64 || c->c2a.cout != 64 || c->c2b.cout != 64 || c->c3a.cout != 128 || c->c3b.cout != 128 || c->c4a.cout != 128 ||
        c->c4b.cout != 128 || c->heads.cout != 512 || c->db.cout != c->ddim)
        return imp_fail(IMP_E_KEY, "state_dict shapes do not match nets/superpoint.py:118-137");
    auto w1 = c->raw.find("conv1a.weight"), b1 = c->raw.find("conv1a.bias"), wp = c->raw.find("convPb.weight"), bp = c->raw.find("convPb.bias");
    if (w1 == c->raw.end() || b1 == c->raw.end() || wp == c->raw.end() || bp == c->raw.end()) return imp_fail(IMP_E_KEY, "missing conv1a / convPb tensors");
    if (w1->second.data.size() != 64 * 9 || b1->second.data.size() != 64 || wp->second.data.size() != 65 * 256 || bp->second.data.size() != 65)
        return imp_fail(IMP_E_KEY, "conv1a / convPb tensors have unexpected sizes");
    {   // convPb as an MFMA 1x1 convolution: 65 output channels padded with zero rows to 128
        HostT wpad, bpad;
        wpad.data.assign((size_t)128 * 256, 0.f);
        bpad.data.assign(128, 0.f);
        memcpy(wpad.data.data(), wp->second.data.data(), (size_t)65 * 256 * sizeof(float));
        memcpy(bpad.data.data(), bp->second.data.data(), 65 * sizeof(float));
        c->raw["convPb_padded.weight"] = std::move(wpad);
        c->raw["convPb_padded.bias"] = std::move(bpad);
        rc = pack_conv(c, {"convPb_padded"}, 256, 1, &c->pb);
        c->raw.erase("convPb_padded.weight");
        c->raw.erase("convPb_padded.bias");
        if (rc) return rc;
        w1 = c->raw.find("conv1a.weight"); b1 = c->raw.find("conv1a.bias"); wp = c->raw.find("convPb.weight"); bp = c->raw.find("convPb.bias");
    }
    std::vector<float> w1t(9 * 64), wpt(256 * 65);
    for (int o = 0; o < 64; ++o)
        for (int t = 0; t < 9; ++t) w1t[t * 64 + o] = w1->second.data[o * 9 + t];
    for (int o = 0; o < 65; ++o)
        for (int k = 0; k < 256; ++k) wpt[k * 65 + o] = wp->second.data[o * 256 + k];
    SP_TRY(hipMalloc(&c->w1a, w1t.size() * 4));
    SP_TRY(hipMemcpy(c->w1a, w1t.data(), w1t.size() * 4, hipMemcpyHostToDevice));
    SP_TRY(hipMalloc(&c->b1a, 64 * 4));
    SP_TRY(hipMemcpy(c->b1a, b1->second.data.data(), 64 * 4, hipMemcpyHostToDevice));
    SP_TRY(hipMalloc(&c->wpb, wpt.size() * 4));
    SP_TRY(hipMemcpy(c->wpb, wpt.data(), wpt.size() * 4, hipMemcpyHostToDevice));
    SP_TRY(hipMalloc(&c->bpb, 65 * 4));
    SP_TRY(hipMemcpy(c->bpb, bp->second.data.data(), 65 * 4, hipMemcpyHostToDevice));
    c->finalized = true;
    return IMP_OK;
}

int imp_sp_detect(imp_sp_ctx* c, const float* image, int B, int H, int W, int nms_radius, float keypoint_threshold, int max_keypoints,
                  int remove_borders, int align_corners, void* stream, int* counts) {
    if (!c || !image || !counts) return imp_fail(IMP_E_ARG, "imp_sp_detect: null argument");
    if (!c->finalized) return imp_fail(IMP_E_STATE, "imp_sp_detect: weights not finalised");
    if (B < 1 || H < 8 || W < 8) return imp_fail(IMP_E_ARG, "imp_sp_detect: image must be at least 8 x 8");
    if (nms_radius < 0 || nms_radius > 8) return imp_fail(IMP_E_ARG, "imp_sp_detect: nms_radius must be in 0..8");
    if (max_keypoints == 0 || max_keypoints < -1) return imp_fail(IMP_E_ARG, "\"max_keypoints\" must be positive or \"-1\"");   // nets/superpoint.py:161-163
    if (max_keypoints > TOPK_CAP) return imp_fail(IMP_E_ARG, "imp_sp_detect: max_keypoints above 16384 is not supported");
    SP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, h = H4 / 2, w = W4 / 2;
    const int Hs = h * 8, Ws = w * 8;
    const size_t needA = (size_t)B * H * W * 64, needB = (size_t)B * H2 * W2 * 64, need_small = (size_t)B * h * w * 512, need_map = (size_t)B * Hs * Ws;
    if (needA > c->capA || needB > c->capB || need_small > c->cap_small || need_map > c->cap_map || (size_t)B * Hs > c->cap_rows || (size_t)B > c->cap_b) {
        SP_TRY(hipDeviceSynchronize());
        free_ws(c);
        SP_TRY(hipMalloc(&c->bufA, std::max(needA, need_small) * 4));
        SP_TRY(hipMalloc(&c->bufB, needB * 4));
        SP_TRY(hipMalloc(&c->dmap, (size_t)B * h * w * 256 * 4));
        SP_TRY(hipMalloc(&c->logits, (size_t)B * h * w * 128 * 4));
        SP_TRY(hipMalloc(&c->scores, need_map * 4));
        SP_TRY(hipMalloc(&c->nms, need_map * 4));
        SP_TRY(hipMalloc(&c->kp0, need_map * 8));
        SP_TRY(hipMalloc(&c->sc0, need_map * 4));
        SP_TRY(hipMalloc(&c->kp1, need_map * 8));
        SP_TRY(hipMalloc(&c->sc1, need_map * 4));
        SP_TRY(hipMalloc(&c->rowcount, (size_t)B * Hs * 4));
        SP_TRY(hipMalloc(&c->rowoff, (size_t)B * Hs * 4));
        SP_TRY(hipMalloc(&c->count0, (size_t)B * 4));
        SP_TRY(hipHostMalloc(&c->count1_host, (size_t)B * 4, hipHostMallocMapped));
        SP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->count1), c->count1_host, 0));
        SP_TRY(hipMalloc(&c->tilecount, (size_t)B * Hs * ((Ws + 31) / 32) * 4));
        SP_TRY(hipMalloc(&c->sel_m, (size_t)B * 4));
        SP_TRY(hipMalloc(&c->selkeys, (size_t)B * TOPK_CAP * 8));
        c->capA = std::max(needA, need_small); c->capB = needB; c->cap_small = need_small; c->cap_map = need_map;
        c->cap_rows = (size_t)B * Hs; c->cap_b = (size_t)B;
    }
    c->detected = false;
    c->B = B; c->H = H; c->W = W; c->h = h; c->w = w;
    c->align_corners = align_corners ? 1 : 0;
    int rc;
    // encoder (nets/superpoint.py:172-183)
    if ((rc = launch_conv_first(c, image, B, H, W, c->bufB, st))) return rc;                     // conv1a + conv1b -> [H2][W2][64]
    if ((rc = launch_conv(c->c2a, c->bufB, 64, 0, B, H2, W2, c->bufA, 64, 1, 0, st))) return rc;
    if ((rc = launch_conv(c->c2b, c->bufA, 64, 0, B, H2, W2, c->bufB, 64, 1, 1, st))) return rc;        // -> [H4][W4][64]
    if ((rc = launch_conv(c->c3a, c->bufB, 64, 0, B, H4, W4, c->bufA, 128, 1, 0, st))) return rc;
    if ((rc = launch_conv(c->c3b, c->bufA, 128, 0, B, H4, W4, c->bufB, 128, 1, 1, st))) return rc;      // -> [h][w][128]
    if ((rc = launch_conv(c->c4a, c->bufB, 128, 0, B, h, w, c->bufA, 128, 1, 0, st))) return rc;
    if ((rc = launch_conv(c->c4b, c->bufA, 128, 0, B, h, w, c->bufB, 128, 1, 0, st))) return rc;
    // both heads' 3x3 convolutions as one launch: channels 0..255 = relu(convPa), 256..511 = relu(convDa)  (:186, :223)
    if ((rc = launch_conv(c->heads, c->bufB, 128, 0, B, h, w, c->bufA, 512, 1, 0, st))) return rc;
    if ((rc = launch_conv(c->db, c->bufA, 512, 256, B, h, w, c->dmap, c->ddim, 0, 0, st))) return rc;   // raw convDb (:224)
    const int npix = B * h * w;
    if ((rc = launch_conv(c->pb, c->bufA, 512, 0, B, h, w, c->logits, 128, 0, 0, st))) return rc;     // convPb logits (:193)
    hipLaunchKernelGGL(sp_softmax_shuffle_kernel, dim3((npix + 3) / 4), dim3(256), 0, st, c->logits, c->scores, npix, h, w);
    SP_TRY(hipGetLastError());
    const bool nms_fast = nms_radius >= 1 && nms_radius <= 6;      // (larger radii: the generic kernel)
    if (nms_fast) {
        switch (nms_radius) {
            case 1: rc = launch_nms_fast<1>(c->scores, c->nms, B, Hs, Ws, keypoint_threshold, remove_borders, c->tilecount, st); break;
            case 2: rc = launch_nms_fast<2>(c->scores, c->nms, B, Hs, Ws, keypoint_threshold, remove_borders, c->tilecount, st); break;
            case 3: rc = launch_nms_fast<3>(c->scores, c->nms, B, Hs, Ws, keypoint_threshold, remove_borders, c->tilecount, st); break;
            case 4: rc = launch_nms_fast<4>(c->scores, c->nms, B, Hs, Ws, keypoint_threshold, remove_borders, c->tilecount, st); break;
            case 5: rc = launch_nms_fast<5>(c->scores, c->nms, B, Hs, Ws, keypoint_threshold, remove_borders, c->tilecount, st); break;
            default: rc = launch_nms_fast<6>(c->scores, c->nms, B, Hs, Ws, keypoint_threshold, remove_borders, c->tilecount, st); break;
        }
        if (rc) return rc;
    } else {
        const int NT = nms_radius <= 6 ? 32 : 16;
        const int E = NT + 10 * nms_radius;
        const size_t lds = (size_t)E * E * (4 * 4 + 1);
        const int tx = (Ws + NT - 1) / NT, ty = (Hs + NT - 1) / NT;
        SP_TRY(imp_grant_dynamic_lds(reinterpret_cast<const void*>(&sp_nms_kernel), lds));
        hipLaunchKernelGGL(sp_nms_kernel, dim3(tx * ty * B), dim3(256), lds, st, c->scores, c->nms, Hs, Ws, nms_radius, NT, tx, ty);
        SP_TRY(hipGetLastError());
    }
    const size_t cap = (size_t)Hs * Ws;
    if (!nms_fast) hipLaunchKernelGGL(sp_rowcount_kernel, dim3(Hs, B), dim3(64), 0, st, c->nms, Hs, Ws, keypoint_threshold, remove_borders, c->rowcount);
    hipLaunchKernelGGL(sp_scan_kernel, dim3(B), dim3(1024), 0, st, c->rowcount, nms_fast ? c->tilecount : nullptr, (Ws + 31) / 32, Hs, c->rowoff, c->count0);
    hipLaunchKernelGGL(sp_compact_kernel, dim3(Hs, B), dim3(64), 0, st, c->nms, Hs, Ws, keypoint_threshold, remove_borders, c->rowoff, c->kp0, c->sc0, cap);
    SP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sp_topk_select_kernel, dim3(B), dim3(1024), 0, st, c->kp0, c->sc0, c->count0, cap, max_keypoints, c->kp1, c->sc1, c->count1,
                       c->selkeys, c->sel_m);
    SP_TRY(hipGetLastError());
    if (max_keypoints >= 0) {
        const int mmax = std::max(max_keypoints, TOPK_DIRECT);
        const size_t lds = (size_t)((mmax + 255) & ~255) * 8;
        SP_TRY(imp_grant_dynamic_lds(reinterpret_cast<const void*>(&sp_topk_rank_kernel), lds));
        hipLaunchKernelGGL(sp_topk_rank_kernel, dim3((mmax + 15) / 16, B), dim3(256), lds, st, c->kp0, cap, max_keypoints, c->selkeys, c->sel_m,
                           c->kp1, c->sc1);
        SP_TRY(hipGetLastError());
    }
    SP_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) counts[b] = c->count1_host[b];
    c->counts.assign(counts, counts + B);
    c->detected = true;
    return IMP_OK;
}

int imp_sp_describe(imp_sp_ctx* c, int b, float* keypoints, float* scores, float* descriptors, void* stream) {
    if (!c) return imp_fail(IMP_E_ARG, "imp_sp_describe: null context");
    if (!c->detected) return imp_fail(IMP_E_STATE, "imp_sp_describe: no imp_sp_detect result");
    if (b < 0 || b >= c->B) return imp_fail(IMP_E_ARG, "imp_sp_describe: image index out of range");
    const int n = c->counts[b];
    if (n == 0) return IMP_OK;
    if (!keypoints || !scores || !descriptors) return imp_fail(IMP_E_ARG, "imp_sp_describe: null output");
    SP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t cap = (size_t)c->h * 8 * c->w * 8;
    hipLaunchKernelGGL(sp_sample_kernel, dim3((n + 3) / 4), dim3(256), 0, st, c->kp1 + (size_t)b * cap * 2, c->sc1 + (size_t)b * cap, n,
                       c->dmap + (size_t)b * c->h * c->w * c->ddim, c->h, c->w, c->ddim, c->align_corners, keypoints, scores, descriptors);
    SP_TRY(hipGetLastError());
    return IMP_OK;
}

int imp_sp_dense(imp_sp_ctx* c, float* scores, float* nms_scores, float* descriptors, void* stream) {
    if (!c) return imp_fail(IMP_E_ARG, "imp_sp_dense: null context");
    if (!c->detected) return imp_fail(IMP_E_STATE, "imp_sp_dense: no imp_sp_detect result");
    SP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nmap = (size_t)c->B * c->h * 8 * c->w * 8;
    if (scores) SP_TRY(hipMemcpyAsync(scores, c->scores, nmap * 4, hipMemcpyDeviceToDevice, st));
    if (nms_scores) SP_TRY(hipMemcpyAsync(nms_scores, c->nms, nmap * 4, hipMemcpyDeviceToDevice, st));
    if (descriptors) {
        const int total = c->B * c->h * c->w;
        hipLaunchKernelGGL(sp_dense_desc_kernel, dim3((total + 3) / 4), dim3(256), 0, st, c->dmap, c->h * c->w, total, c->ddim, descriptors);
        SP_TRY(hipGetLastError());
    }
    return IMP_OK;
}

int imp_sp_op_conv(imp_sp_ctx* c, int layer, const float* in, int B, int H, int W, float* out, int relu, int pool, void* stream) {
    if (!c || !in || !out) return imp_fail(IMP_E_ARG, "imp_sp_op_conv: null argument");
    if (!c->finalized) return imp_fail(IMP_E_STATE, "imp_sp_op_conv: weights not finalised");
    if (B < 1 || H < 1 || W < 1) return imp_fail(IMP_E_ARG, "imp_sp_op_conv: bad shape");
    SP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (layer == 0) {
        const size_t nthr = (size_t)B * H * W * 16;
        hipLaunchKernelGGL(sp_conv1a_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, in, c->w1a, c->b1a, out, B, H, W);
        SP_TRY(hipGetLastError());
        return IMP_OK;
    }
    const ConvW* ws[] = {nullptr, &c->c1b, &c->c2a, &c->c2b, &c->c3a, &c->c3b, &c->c4a, &c->c4b, &c->heads, &c->db};
    if (layer < 1 || layer > 9) return imp_fail(IMP_E_ARG, "imp_sp_op_conv: layer must be 0..9");
    const ConvW& w = *ws[layer];
    if (pool && w.taps != 9) return imp_fail(IMP_E_ARG, "imp_sp_op_conv: pooling is fused into the 3x3 kernels only");
    return launch_conv(w, in, w.cin, 0, B, H, W, out, w.cout, relu, pool, st);
}

}  // extern "C"
