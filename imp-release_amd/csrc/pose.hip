// Pose step of the iterative loops on the GPU, gfx950:
//   estimate_pose              eval/pose_estimation.py:92-115     essential matrix from the current matches
//   decompose_essential_mat    eval/pose_estimation.py:13-89      4-way cheirality vote (R1 | R2, +-t)
//
// The reference parks the GPU during every pose estimate (cv2.findEssentialMat(USAC_MAGSAC) on the host, 7 times per pair,
// eval/matching.py:84-87).  Here the estimate is a batch of small kernels:
//   * H seeded minimal samples in parallel, one thread each: FIVE-POINT solver (round 3, default; pose_fivept.h: up to 10 essential
//     matrices per sample, like the minimal solver inside cv2.findEssentialMat) or the linear eight-point solver of round 2 (Hartley
//     conditioning, null vector by Gauss-Jordan, projection onto the essential manifold);
//   * one workgroup per candidate model: MAGSAC++ sigma-marginalised quality (round 3, default) or the Sampson inlier count;
//   * ONE workgroup picks the first best model and refines it by (weighted) least squares - IRLS with the MAGSAC++ weights, or refits
//     on the consensus set - up to 3 times while not worse, then decomposes E;
//   * the 4-way cheirality vote with per-point DLT triangulation, and the masks (the reference's all-True-outside-the-consensus mask and
//     the geometric one).
// Everything in fp64 (n <= a few thousand correspondences).
//
// PARITY: the solver follows the PUBLISHED algorithms behind the reference's call (five-point minimal solver, MAGSAC++ quality and IRLS)
// but it is NOT OpenCV's implementation - a third-party randomized solver with its own sampler, termination and local optimisation, no
// golden vectors in the reference, and cv2 is absent from the build image: parity with it is unpinned and not claimed.  What IS pinned:
// these kernels against their CPU twin oracle/pose_oracle.py (same samples, same algebra, every mode), the five-point header against the
// twin on the host (tests/test_pose.py), the cheirality vote against the geometric definition, recovery of known poses.
#include "imp_kernels.h"
#include "../../include/imp_hip.h"
#include "pose_fivept.h"
#include <mutex>
#include <vector>

namespace {

__host__ __device__ inline unsigned pose_rand(unsigned seed, unsigned h, unsigned k) {
    unsigned x = seed * 0x9E3779B1u + h * 0x85EBCA77u + k * 0xC2B2AE3Du + 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

// cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (a is destroyed: eigenvalues on its diagonal); v = eigenvectors (columns)
template <int N>
__device__ void jacobi_eig(double (&a)[N][N], double (&v)[N][N]) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < N; ++i) {
            diag += a[i][i] * a[i][i];
            for (int j = i + 1; j < N; ++j) off += a[i][j] * a[i][j];
        }
        if (off <= 1e-26 * diag || off == 0.0) break;     // off-diagonal norm below 1e-13 of the diagonal: converged in fp64
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// the same cyclic Jacobi for a 9 x 9 matrix held in LDS, executed by ONE wave: lane k owns row / column k of every rotation, the
// rotation angle is computed redundantly by all lanes (a single thread walking a 9 x 9 array in scratch memory takes ~1 ms)
__device__ void jacobi9_wave(double* A, double* V, int lane) {
    auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (int i = lane; i < 81; i += 64) V[i] = (i / 9 == i % 9) ? 1.0 : 0.0;
    sync();
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = lane; i < 81; i += 64) {
            const int r = i / 9, c = i % 9;
            const double x = A[i];
            if (r == c) diag += x * x; else if (r < c) off += x * x;
        }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); diag += __shfl_xor(diag, o); }
        if (off <= 1e-26 * diag || off == 0.0) break;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = A[p * 9 + q];
                if (apq == 0.0) continue;                         // uniform
                const double theta = (A[q * 9 + q] - A[p * 9 + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                sync();
                if (lane < 9) {
                    const double akp = A[lane * 9 + p], akq = A[lane * 9 + q];
                    A[lane * 9 + p] = c * akp - s * akq;
                    A[lane * 9 + q] = s * akp + c * akq;
                    const double vkp = V[lane * 9 + p], vkq = V[lane * 9 + q];
                    V[lane * 9 + p] = c * vkp - s * vkq;
                    V[lane * 9 + q] = s * vkp + c * vkq;
                }
                sync();
                if (lane < 9) {
                    const double apk = A[p * 9 + lane], aqk = A[q * 9 + lane];
                    A[p * 9 + lane] = c * apk - s * aqk;
                    A[q * 9 + lane] = s * apk + c * aqk;
                }
                sync();
            }
    }
}

// SVD of a 3x3 matrix through the eigen-decomposition of E^T E: singular values descending, U, V with u2 = u0 x u1, v2 = v0 x v1
__device__ void svd3(const double (&E)[3][3], double (&U)[3][3], double (&s)[3], double (&V)[3][3]) {
    double a[3][3], ev[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = E[0][i] * E[0][j] + E[1][i] * E[1][j] + E[2][i] * E[2][j];
    jacobi_eig<3>(a, ev);
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (a[idx[j]][idx[j]] > a[idx[i]][idx[i]]) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    for (int c = 0; c < 2; ++c) {
        s[c] = sqrt(fmax(a[idx[c]][idx[c]], 0.0));
        for (int r = 0; r < 3; ++r) V[r][c] = ev[r][idx[c]];
        for (int r = 0; r < 3; ++r) U[r][c] = (E[r][0] * V[0][c] + E[r][1] * V[1][c] + E[r][2] * V[2][c]) / fmax(s[c], 1e-300);
    }
    s[2] = sqrt(fmax(a[idx[2]][idx[2]], 0.0));
    V[0][2] = V[1][0] * V[2][1] - V[2][0] * V[1][1];
    V[1][2] = V[2][0] * V[0][1] - V[0][0] * V[2][1];
    V[2][2] = V[0][0] * V[1][1] - V[1][0] * V[0][1];
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
}

// conditioned fundamental estimate F -> essential matrix: un-conditioning E = T1^T F T0, projection onto singular values (1, 1, 0)
__device__ bool essential_from_F(const double (&F)[3][3], const double (&T0)[3], const double (&T1)[3], double (&Eo)[3][3]) {
    // T = [[s, 0, -s cx], [0, s, -s cy], [0, 0, 1]] stored as (s, cx, cy)
    double FT0[3][3];
    for (int r = 0; r < 3; ++r) {
        FT0[r][0] = F[r][0] * T0[0];
        FT0[r][1] = F[r][1] * T0[0];
        FT0[r][2] = -F[r][0] * T0[0] * T0[1] - F[r][1] * T0[0] * T0[2] + F[r][2];
    }
    double E[3][3];
    for (int c = 0; c < 3; ++c) {
        E[0][c] = T1[0] * FT0[0][c];
        E[1][c] = T1[0] * FT0[1][c];
        E[2][c] = -T1[0] * T1[1] * FT0[0][c] - T1[0] * T1[2] * FT0[1][c] + FT0[2][c];
    }
    double U[3][3], s[3], V[3][3];
    svd3(E, U, s, V);
    if (!(s[1] > 1e-12 * s[0]) || !(s[0] > 0.0)) return false;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Eo[r][c] = U[r][0] * V[c][0] + U[r][1] * V[c][1];
    return true;
}
// least-squares form: smallest eigenvector of the 9x9 normal matrix of conditioned correspondences
__device__ bool essential_from_normal(double (&ata)[9][9], const double (&T0)[3], const double (&T1)[3], double (&Eo)[3][3]) {
    double v[9][9];
    jacobi_eig<9>(ata, v);
    int m = 0;
    for (int i = 1; i < 9; ++i) if (ata[i][i] < ata[m][m]) m = i;
    double F[3][3];
    for (int i = 0; i < 9; ++i) F[i / 3][i % 3] = v[i][m];
    return essential_from_F(F, T0, T1, Eo);
}

__device__ __forceinline__ double sampson_sq(const double* E, double x0, double y0, double x1, double y1) {
    const double a0 = E[0] * x0 + E[1] * y0 + E[2], a1 = E[3] * x0 + E[4] * y0 + E[5], a2 = E[6] * x0 + E[7] * y0 + E[8];
    const double b0 = E[0] * x1 + E[3] * y1 + E[6], b1 = E[1] * x1 + E[4] * y1 + E[7];
    const double num = x1 * a0 + y1 * a1 + a2;
    return num * num / fmax(a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1, 1e-30);
}

// MAGSAC++ (Barath et al., CVPR 2020) sigma-marginalised weight of a residual: the noise scale sigma is unknown, uniform on (0, sigma_max];
// residuals are chi-distributed with nu = 4 degrees of freedom (the paper's choice for epipolar geometry with the Sampson distance), a point
// is an inlier of scale sigma while r < k sigma, k = 3.64 (0.99 quantile).  Marginalising the inlier likelihood over sigma gives
//     w(r) ~ Gamma_u((nu - 1) / 2, r^2 / (2 sigma_max^2)) - Gamma_u((nu - 1) / 2, k^2 / 2)     for r < k sigma_max, else 0
// with the upper incomplete gamma function Gamma_u(3/2, x) = sqrt(pi) / 2 erfc(sqrt x) + sqrt x e^-x; normalised here to w(0) = 1.
// The quality of a model is sum_i w(r_i) and its refinement is a least-squares fit weighted by w (the paper's IRLS step).
constexpr double MAGSAC_K = 3.64;
__host__ __device__ inline double gamma_u_3_2(double x) {
    const double rx = sqrt(x);
    return 0.88622692545275801 * erfc(rx) + rx * exp(-x);
}
__host__ __device__ inline double magsac_weight(double r2, double sigma_max2) {
    if (!(r2 < MAGSAC_K * MAGSAC_K * sigma_max2)) return 0.0;
    const double gk = gamma_u_3_2(0.5 * MAGSAC_K * MAGSAC_K);
    return (gamma_u_3_2(0.5 * r2 / sigma_max2) - gk) / (0.88622692545275801 - gk);
}
constexpr double QUALITY_SCALE = 4096.0;     // qualities are compared as integers floor(Q x 4096): ties resolve to the lowest hypothesis index

// one thread per hypothesis
__global__ __launch_bounds__(64) void pose_hypotheses_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n,
                                                             int H, unsigned seed, double* __restrict__ Eh, int* __restrict__ valid) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    int ids[8];
    bool ok = true;
    for (int k = 0; k < 8; ++k) {
        ids[k] = (int)(pose_rand(seed, (unsigned)h, (unsigned)k) % (unsigned)n);
        for (int j = 0; j < k; ++j) ok &= ids[j] != ids[k];
    }
    double E[3][3];
    if (ok) {
        double T[2][3];
        double ax[8], ay[8], bx[8], by[8];
        for (int v = 0; v < 2; ++v) {
            double cx = 0, cy = 0;
            for (int k = 0; k < 8; ++k) { const double2 p = v ? x1[ids[k]] : x0[ids[k]]; cx += p.x; cy += p.y; }
            cx /= 8.0; cy /= 8.0;
            double md = 0;
            for (int k = 0; k < 8; ++k) { const double2 p = v ? x1[ids[k]] : x0[ids[k]]; md += sqrt((p.x - cx) * (p.x - cx) + (p.y - cy) * (p.y - cy)); }
            const double s = 1.4142135623730951 / fmax(md / 8.0, 1e-12);
            T[v][0] = s; T[v][1] = cx; T[v][2] = cy;
            for (int k = 0; k < 8; ++k) {
                const double2 p = v ? x1[ids[k]] : x0[ids[k]];
                (v ? bx : ax)[k] = (p.x - cx) * s;
                (v ? by : ay)[k] = (p.y - cy) * s;
            }
        }
        // minimal sample: the 8 x 9 system has a one-dimensional null space - Gauss-Jordan with full pivoting gives it directly
        // (the same vector, up to scale, as the smallest eigenvector of A^T A that the least-squares refit uses)
        double A[8][9];
        for (int k = 0; k < 8; ++k) {
            const double r[9] = {bx[k] * ax[k], bx[k] * ay[k], bx[k], by[k] * ax[k], by[k] * ay[k], by[k], ax[k], ay[k], 1.0};
            for (int j = 0; j < 9; ++j) A[k][j] = r[j];
        }
        int pcol[8];
        unsigned used = 0;
        for (int r = 0; r < 8 && ok; ++r) {
            int pi = r, pj = -1;
            double best = 0.0;
            for (int i = r; i < 8; ++i)
                for (int j = 0; j < 9; ++j)
                    if (!((used >> j) & 1u) && fabs(A[i][j]) > best) { best = fabs(A[i][j]); pi = i; pj = j; }
            if (pj < 0 || best < 1e-12) { ok = false; break; }
            if (pi != r) for (int j = 0; j < 9; ++j) { const double t = A[r][j]; A[r][j] = A[pi][j]; A[pi][j] = t; }
            used |= 1u << pj;
            pcol[r] = pj;
            const double inv = 1.0 / A[r][pj];
            for (int j = 0; j < 9; ++j) A[r][j] *= inv;
            for (int i = 0; i < 8; ++i)
                if (i != r) {
                    const double f = A[i][pj];
                    if (f != 0.0) for (int j = 0; j < 9; ++j) A[i][j] -= f * A[r][j];
                }
        }
        if (ok) {
            int q = 0;
            while ((used >> q) & 1u) ++q;                    // the free column
            double fvec[9];
            fvec[q] = 1.0;
            for (int r = 0; r < 8; ++r) fvec[pcol[r]] = -A[r][q];
            double F[3][3];
            for (int i = 0; i < 9; ++i) F[i / 3][i % 3] = fvec[i];
            ok = essential_from_F(F, T[0], T[1], E);
        }
    }
    valid[h] = ok ? 1 : 0;
    if (ok) for (int i = 0; i < 9; ++i) Eh[(long)h * 9 + i] = E[i / 3][i % 3];
}

// five-point sampler: one thread per minimal sample, up to 10 models each (pose_fivept.h); candidate c of sample h lives at slot 10 h + c.
// The solver is one long dependent chain of run-time-indexed array accesses: with the arrays in scratch memory (an L2 round trip each) a
// call took 1.84 ms however few samples it had; every thread gets its Work in LDS instead (8 threads per workgroup = 47 KB)
constexpr int FP_THREADS = 8;
__global__ __launch_bounds__(64) void pose_hypotheses5_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n,
                                                              int H, unsigned seed, double* __restrict__ Eh, int* __restrict__ valid) {
    __shared__ fivept::Work work[FP_THREADS];
    if (threadIdx.x >= FP_THREADS) return;
    const int h = blockIdx.x * FP_THREADS + threadIdx.x;
    if (h >= H) return;
    int ids[5];
    bool ok = true;
    for (int k = 0; k < 5; ++k) {
        ids[k] = (int)(pose_rand(seed, (unsigned)h, (unsigned)k) % (unsigned)n);
        for (int j = 0; j < k; ++j) ok &= ids[j] != ids[k];
    }
    int nsol = 0;
    double* Eo = Eh + (long)h * 90;
    if (ok) {
        double a[5][2], b[5][2];
        for (int k = 0; k < 5; ++k) { a[k][0] = x0[ids[k]].x; a[k][1] = x0[ids[k]].y; b[k][0] = x1[ids[k]].x; b[k][1] = x1[ids[k]].y; }
        nsol = fivept::five_point(a, b, Eo, work[threadIdx.x]);
    }
    for (int c = 0; c < 10; ++c) valid[(long)h * 10 + c] = c < nsol ? 1 : 0;
}

// one workgroup per hypothesis: inlier count (magsac == 0) or sigma-marginalised quality floor(4096 sum_i w(r_i)) (magsac != 0)
__global__ __launch_bounds__(256) void pose_score_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n,
                                                         const double* __restrict__ Eh, const int* __restrict__ valid, double thr2, int magsac,
                                                         int* __restrict__ counts) {
    const int h = blockIdx.x;
    __shared__ double red[4];
    double c = 0;
    if (valid[h]) {
        double E[9];
        for (int i = 0; i < 9; ++i) E[i] = Eh[(long)h * 9 + i];
        for (int i = threadIdx.x; i < n; i += 256) {
            const double r2 = sampson_sq(E, x0[i].x, x0[i].y, x1[i].x, x1[i].y);
            c += magsac ? magsac_weight(r2, thr2) : (r2 < thr2 ? 1.0 : 0.0);
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double q = (red[0] + red[1]) + (red[2] + red[3]);
        counts[h] = valid[h] ? (magsac ? (int)floor(q * QUALITY_SCALE) : (int)q) : -1;
    }
}

__device__ double block_sum(double v, double* sm) {        // 1024 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < 16; ++w) t += sm[w];
    return t;
}

// one workgroup: first best hypothesis -> consensus refits -> decomposition of E
__global__ __launch_bounds__(1024) void pose_consensus_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n, int H,
                                                           const double* __restrict__ Eh, const int* __restrict__ counts, double thr2, int magsac, int nsample,
                                                           unsigned char* __restrict__ inl, double* __restrict__ out) {
    // out: [0..8] E, [9..17] R, [18..20] t, [21] inliers of E, [22] cheirality inliers, [23] ok flag, [24..32] R1, [33..41] R2, [42..44] t
    __shared__ double sm[16 * 45];
    __shared__ double Es[9], Et[9];
    __shared__ int s_best, s_cnt, s_ok;
    const int tid = threadIdx.x;
    {   // first best hypothesis (largest count, lowest index): strided scan + wave / workgroup reduction
        int best = -1, bi = 0x7fffffff;
        for (int h = tid; h < H; h += 1024) { const int c = counts[h]; if (c > best) { best = c; bi = h; } }
        for (int o = 32; o > 0; o >>= 1) {
            const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        int* smi = reinterpret_cast<int*>(sm);
        if ((tid & 63) == 0) { smi[2 * (tid >> 6)] = best; smi[2 * (tid >> 6) + 1] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w)
                if (smi[2 * w] > best || (smi[2 * w] == best && smi[2 * w + 1] < bi)) { best = smi[2 * w]; bi = smi[2 * w + 1]; }
            s_best = best >= 0 ? bi : -1; s_cnt = best;
            if (best >= 0) for (int i = 0; i < 9; ++i) Es[i] = Eh[(long)bi * 9 + i];
        }
        __syncthreads();
    }
    // nsample = size of the minimal sample (5 or 8): a model has to explain at least that many matches (half of it in weight units)
    if (s_best < 0 || s_cnt < (magsac ? (int)(nsample * QUALITY_SCALE / 2) : nsample)) { if (tid == 0) out[23] = 0.0; return; }
    // per-point weight of the current model: 0 / 1 membership of the consensus set, or the sigma-marginalised weight (IRLS)
    auto weight = [&](const double* E, int i) {
        const double r2 = sampson_sq(E, x0[i].x, x0[i].y, x1[i].x, x1[i].y);
        return magsac ? magsac_weight(r2, thr2) : (r2 < thr2 ? 1.0 : 0.0);
    };
    for (int i = tid; i < n; i += 1024) inl[i] = sampson_sq(Es, x0[i].x, x0[i].y, x1[i].x, x1[i].y) < thr2;
    __syncthreads();
    for (int round = 0; round < (n >= 8 ? 3 : 0); ++round) {      // the linear refit needs 8 correspondences
        // conditioning of the (weighted) consensus set
        double c = 0, sx0 = 0, sy0 = 0, sx1 = 0, sy1 = 0;
        for (int i = tid; i < n; i += 1024) { const double w = weight(Es, i); if (w > 0) { c += w; sx0 += w * x0[i].x; sy0 += w * x0[i].y; sx1 += w * x1[i].x; sy1 += w * x1[i].y; } }
        const double cnt = block_sum(c, sm);
        const double cx0 = block_sum(sx0, sm) / cnt, cy0 = block_sum(sy0, sm) / cnt, cx1 = block_sum(sx1, sm) / cnt, cy1 = block_sum(sy1, sm) / cnt;
        double d0 = 0, d1 = 0;
        for (int i = tid; i < n; i += 1024) { const double w = weight(Es, i); if (w > 0) {
            d0 += w * sqrt((x0[i].x - cx0) * (x0[i].x - cx0) + (x0[i].y - cy0) * (x0[i].y - cy0));
            d1 += w * sqrt((x1[i].x - cx1) * (x1[i].x - cx1) + (x1[i].y - cy1) * (x1[i].y - cy1));
        } }
        const double s0 = 1.4142135623730951 / fmax(block_sum(d0, sm) / cnt, 1e-12), s1 = 1.4142135623730951 / fmax(block_sum(d1, sm) / cnt, 1e-12);
        double acc[45];
        for (int k = 0; k < 45; ++k) acc[k] = 0.0;
        for (int i = tid; i < n; i += 1024) { const double w = weight(Es, i); if (w > 0) {
            const double ax = (x0[i].x - cx0) * s0, ay = (x0[i].y - cy0) * s0, bx = (x1[i].x - cx1) * s1, by = (x1[i].y - cy1) * s1;
            const double r[9] = {bx * ax, bx * ay, bx, by * ax, by * ay, by, ax, ay, 1.0};
            int k = 0;
            for (int a = 0; a < 9; ++a) for (int b2 = a; b2 < 9; ++b2) acc[k++] += w * r[a] * r[b2];
        } }
        for (int k = 0; k < 45; ++k) for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
        __syncthreads();
        if ((tid & 63) == 0) for (int k = 0; k < 45; ++k) sm[(tid >> 6) * 45 + k] = acc[k];
        __syncthreads();
        __shared__ double JA[81], JV[81];
        if (tid < 45) {                                            // assemble the symmetric normal matrix (fixed summation order)
            int a = 0, rem = tid;
            while (rem >= 9 - a) { rem -= 9 - a; ++a; }
            const int b2 = a + rem;
            double t = 0;
            for (int w = 0; w < 16; ++w) t += sm[w * 45 + tid];
            JA[a * 9 + b2] = JA[b2 * 9 + a] = t;
        }
        __syncthreads();
        if (tid < 64) jacobi9_wave(JA, JV, tid);
        __syncthreads();
        if (tid == 0) {
            int m = 0;
            for (int i = 1; i < 9; ++i) if (JA[i * 9 + i] < JA[m * 9 + m]) m = i;
            double F[3][3];
            for (int i = 0; i < 9; ++i) F[i / 3][i % 3] = JV[i * 9 + m];
            const double T0[3] = {s0, cx0, cy0}, T1[3] = {s1, cx1, cy1};
            double E2[3][3];
            s_ok = essential_from_F(F, T0, T1, E2) ? 1 : 0;
            if (s_ok) for (int i = 0; i < 9; ++i) Et[i] = E2[i / 3][i % 3];
        }
        __syncthreads();
        if (!s_ok) break;
        double c2 = 0;
        for (int i = tid; i < n; i += 1024) c2 += weight(Et, i);
        const double q2 = block_sum(c2, sm);
        const int cnt2 = magsac ? (int)floor(q2 * QUALITY_SCALE) : (int)q2;
        if (cnt2 < s_cnt) break;                                   // uniform: kept only while not worse
        const bool same = cnt2 == s_cnt;
        __syncthreads();
        if (tid == 0) { s_cnt = cnt2; for (int i = 0; i < 9; ++i) Es[i] = Et[i]; }
        __syncthreads();
        for (int i = tid; i < n; i += 1024) inl[i] = sampson_sq(Es, x0[i].x, x0[i].y, x1[i].x, x1[i].y) < thr2;
        __syncthreads();
        if (same && round > 0) break;
    }
    // decomposition (cv2.decomposeEssentialMat): U, V^T with positive determinant, R1 = U W V^T, R2 = U W^T V^T, t = u2
    if (tid == 0) {
        double E[3][3], U[3][3], s[3], V[3][3];
        for (int i = 0; i < 9; ++i) E[i / 3][i % 3] = Es[i];
        svd3(E, U, s, V);            // u2 = u0 x u1 and v2 = v0 x v1: both determinants are +1 by construction
        // W = [[0,1,0],[-1,0,0],[0,0,1]]:  U W = [-u1, u0, u2],  U W^T = [u1, -u0, u2]
        for (int r = 0; r < 3; ++r)
            for (int c2 = 0; c2 < 3; ++c2) {
                out[24 + r * 3 + c2] = -U[r][1] * V[c2][0] + U[r][0] * V[c2][1] + U[r][2] * V[c2][2];
                out[33 + r * 3 + c2] = U[r][1] * V[c2][0] - U[r][0] * V[c2][1] + U[r][2] * V[c2][2];
            }
        for (int r = 0; r < 3; ++r) out[42 + r] = U[r][2];
        for (int i = 0; i < 9; ++i) out[i] = Es[i];
        out[21] = (double)s_cnt; out[23] = 1.0;                    // (magsac: [21] holds the quality x 4096; the inlier count follows from the mask)
    }
}

// cheirality test (eval/pose_estimation.py:14-27,40-68) of every inlier against the four candidates (R1,t), (R2,t), (R1,-t), (R2,-t):
// bit k of bits[i] = "point i is in front of both cameras of candidate k"; good[k] counts them (integer atomics: order-free)
__global__ __launch_bounds__(256) void pose_cheirality_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n,
                                                              const double* __restrict__ out, const unsigned char* __restrict__ inl,
                                                              double dist_thresh, unsigned char* __restrict__ bits, int* __restrict__ good) {
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (out[23] != 0.0 && i < n && inl[i]) {
        unsigned char bb = 0;
        for (int k = 0; k < 4; ++k) {
            const double* R = out + 24 + (k & 1) * 9;
            const double sg = k >= 2 ? -1.0 : 1.0;
            const double P[3][4] = {{R[0], R[1], R[2], sg * out[42]}, {R[3], R[4], R[5], sg * out[43]}, {R[6], R[7], R[8], sg * out[44]}};
            // DLT rows (cv2.triangulatePoints): x P0[2] - P0[0], y P0[2] - P0[1] with P0 = [I | 0], and the same for P
            const double A[4][4] = {{-1, 0, x0[i].x, 0}, {0, -1, x0[i].y, 0},
                                    {x1[i].x * P[2][0] - P[0][0], x1[i].x * P[2][1] - P[0][1], x1[i].x * P[2][2] - P[0][2], x1[i].x * P[2][3] - P[0][3]},
                                    {x1[i].y * P[2][0] - P[1][0], x1[i].y * P[2][1] - P[1][1], x1[i].y * P[2][2] - P[1][2], x1[i].y * P[2][3] - P[1][3]}};
            double ata[4][4], ev[4][4];
            for (int a = 0; a < 4; ++a) for (int b2 = 0; b2 < 4; ++b2) ata[a][b2] = A[0][a] * A[0][b2] + A[1][a] * A[1][b2] + A[2][a] * A[2][b2] + A[3][a] * A[3][b2];
            jacobi_eig<4>(ata, ev);
            int m = 0;
            for (int a = 1; a < 4; ++a) if (ata[a][a] < ata[m][m]) m = a;
            const double Q[4] = {ev[0][m], ev[1][m], ev[2][m], ev[3][m]};
            bool ok = Q[2] * Q[3] > 0;
            const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
            ok = ok && Z < dist_thresh;
            const double zc = P[2][0] * X + P[2][1] * Y + P[2][2] * Z + P[2][3];
            ok = ok && zc > 0 && zc < dist_thresh;
            if (ok) { bb |= 1u << k; atomicAdd(&cnt[k], 1); }
        }
        bits[i] = bb;
    }
    __syncthreads();
    if (threadIdx.x < 4 && cnt[threadIdx.x]) atomicAdd(&good[threadIdx.x], cnt[threadIdx.x]);
}

// the vote: most points in front, first candidate on ties (eval/pose_estimation.py:80-89).  Two masks:
//   inl     the geometric one: in the consensus of E AND in front of both cameras
//   refmask the one the reference returns (:113-114): `mask = E_mask.ravel() >= 0` is all True, then only the consensus entries are
//           overwritten with the cheirality result - matches OUTSIDE the consensus stay True.  The loops take their inlier ratio and
//           their early-exit indices from this mask (eval/matching.py:89-90,113), so the drop-in has to reproduce it
__global__ __launch_bounds__(256) void pose_vote_kernel(int n, const int* __restrict__ good, const unsigned char* __restrict__ bits,
                                                        unsigned char* __restrict__ inl, unsigned char* __restrict__ refmask, double* __restrict__ out) {
    if (out[23] == 0.0) return;
    const int best = max(max(good[0], good[1]), max(good[2], good[3]));
    const int k = good[0] == best ? 0 : (good[1] == best ? 1 : (good[2] == best ? 2 : 3));
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const unsigned char front = (bits[i] >> k) & 1;
        refmask[i] = inl[i] ? front : 1;
        if (inl[i]) inl[i] = front;
    }
    if (i == 0) {
        for (int j = 0; j < 9; ++j) out[9 + j] = out[24 + (k & 1) * 9 + j];
        for (int j = 0; j < 3; ++j) out[18 + j] = (k >= 2 ? -1.0 : 1.0) * out[42 + j];
        out[22] = (double)best;
    }
}

struct PoseWs {
    int device = -1;
    size_t cap_n = 0, cap_h = 0;
    double2 *x0 = nullptr, *x1 = nullptr;
    double *Eh = nullptr, *out = nullptr;
    int *valid = nullptr, *counts = nullptr, *good = nullptr;
    unsigned char *inl = nullptr, *bits = nullptr, *refmask = nullptr;
};

}  // namespace

extern "C" int imp_estimate_pose(const float* kpts0, const float* kpts1, int n, const double* K0, const double* K1, double norm_thresh,
                                 int iterations, unsigned seed, int device, double* E, double* R, double* t, unsigned char* mask,
                                 unsigned char* consensus, int* n_inliers, int flags, void* stream) {
    if (!kpts0 || !kpts1 || !K0 || !K1 || !E || !R || !t || !mask || !n_inliers || iterations < 1) return IMP_E_ARG;
    *n_inliers = 0;
    const bool eight = (flags & 2) != 0;                   // IMP_POSE_8PT: the linear eight-point sampler of round 2
    if (n < (eight ? 8 : 5)) return 1;                     // eval/pose_estimation.py:93: None below 5 matches (8 for the eight-point sampler)
    if (hipSetDevice(device) != hipSuccess) return IMP_E_HIP;
    hipStream_t st = (hipStream_t)stream;
    static thread_local PoseWs ws;
    if (ws.device != device || (size_t)n > ws.cap_n || (size_t)iterations > ws.cap_h) {
        for (void* p : {(void*)ws.x0, (void*)ws.x1, (void*)ws.Eh, (void*)ws.out, (void*)ws.valid, (void*)ws.counts, (void*)ws.inl, (void*)ws.good, (void*)ws.bits, (void*)ws.refmask})
            if (p) (void)hipFree(p);
        ws = PoseWs();
        const size_t cn = (size_t)n < 4096 ? 4096 : (size_t)n, ch = (size_t)iterations < 2048 ? 2048 : (size_t)iterations;
        if (hipMalloc(&ws.x0, cn * sizeof(double2)) != hipSuccess || hipMalloc(&ws.x1, cn * sizeof(double2)) != hipSuccess ||
            hipMalloc(&ws.Eh, ch * 10 * 9 * sizeof(double)) != hipSuccess || hipMalloc(&ws.out, 48 * sizeof(double)) != hipSuccess ||
            hipMalloc(&ws.good, 4 * sizeof(int)) != hipSuccess || hipMalloc(&ws.bits, cn) != hipSuccess ||
            hipMalloc(&ws.valid, ch * 10 * sizeof(int)) != hipSuccess || hipMalloc(&ws.counts, ch * 10 * sizeof(int)) != hipSuccess ||
            hipMalloc(&ws.inl, cn) != hipSuccess || hipMalloc(&ws.refmask, cn) != hipSuccess)
            return IMP_E_NOMEM;
        ws.device = device; ws.cap_n = cn; ws.cap_h = ch;
    }
    // normalised coordinates on the host (n is small): (x - cx) / fx, (y - cy) / fy per camera
    std::vector<double2> h0(n), h1(n);
    for (int i = 0; i < n; ++i) {
        h0[i] = double2{((double)kpts0[2 * i] - K0[2]) / K0[0], ((double)kpts0[2 * i + 1] - K0[5]) / K0[4]};
        h1[i] = double2{((double)kpts1[2 * i] - K1[2]) / K1[0], ((double)kpts1[2 * i + 1] - K1[5]) / K1[4]};
    }
    if (hipMemcpyAsync(ws.x0, h0.data(), n * sizeof(double2), hipMemcpyHostToDevice, st) != hipSuccess) return IMP_E_HIP;
    if (hipMemcpyAsync(ws.x1, h1.data(), n * sizeof(double2), hipMemcpyHostToDevice, st) != hipSuccess) return IMP_E_HIP;
    const double thr = norm_thresh / ((K0[0] + K0[4] + K1[0] + K1[4]) / 4.0);
    const int ncand = eight ? iterations : iterations * 10;
    if (eight) hipLaunchKernelGGL(pose_hypotheses_kernel, dim3((iterations + 63) / 64), dim3(64), 0, st, ws.x0, ws.x1, n, iterations, seed, ws.Eh, ws.valid);
    else hipLaunchKernelGGL(pose_hypotheses5_kernel, dim3((iterations + FP_THREADS - 1) / FP_THREADS), dim3(64), 0, st, ws.x0, ws.x1, n, iterations, seed, ws.Eh, ws.valid);
    hipLaunchKernelGGL(pose_score_kernel, dim3(ncand), dim3(256), 0, st, ws.x0, ws.x1, n, ws.Eh, ws.valid, thr * thr, flags & 1, ws.counts);
    // the cheirality step of the reference normalises with K = (K0 + K1) / 2 (eval/pose_estimation.py:29-33): with K0 == K1 (every
    // caller in the repo) these are the coordinates above; a caller with two different cameras gets per-camera normalisation
    hipLaunchKernelGGL(pose_consensus_kernel, dim3(1), dim3(1024), 0, st, ws.x0, ws.x1, n, ncand, ws.Eh, ws.counts, thr * thr, flags & 1, eight ? 8 : 5, ws.inl, ws.out);
    if (hipMemsetAsync(ws.good, 0, 4 * sizeof(int), st) != hipSuccess) return IMP_E_HIP;
    hipLaunchKernelGGL(pose_cheirality_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws.x0, ws.x1, n, ws.out, ws.inl, 1000.0, ws.bits, ws.good);
    hipLaunchKernelGGL(pose_vote_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, ws.good, ws.bits, ws.inl, ws.refmask, ws.out);
    double out[24];
    if (hipMemcpyAsync(out, ws.out, sizeof out, hipMemcpyDeviceToHost, st) != hipSuccess) return IMP_E_HIP;
    if (hipMemcpyAsync(mask, ws.refmask, n, hipMemcpyDeviceToHost, st) != hipSuccess) return IMP_E_HIP;
    if (consensus && hipMemcpyAsync(consensus, ws.inl, n, hipMemcpyDeviceToHost, st) != hipSuccess) return IMP_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return IMP_E_HIP;
    if (hipGetLastError() != hipSuccess) return IMP_E_HIP;
    if (out[23] == 0.0) return 1;
    for (int i = 0; i < 9; ++i) { E[i] = out[i]; R[i] = out[9 + i]; }
    for (int i = 0; i < 3; ++i) t[i] = out[18 + i];
    *n_inliers = (int)out[22];
    return IMP_OK;
}
