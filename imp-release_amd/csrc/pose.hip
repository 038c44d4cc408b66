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
#include <atomic>
#include "../../include/imp_hip.h"
#include "pose_fivept.h"
#include <cstdlib>
#include <cstring>

namespace {

__host__ __device__ inline unsigned pose_rand(unsigned seed, unsigned h, unsigned k) {
    unsigned x = seed * 0x9E3779B1u + h * 0x85EBCA77u + k * 0xC2B2AE3Du + 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

// cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (a is destroyed: eigenvalues on its diagonal); v = eigenvectors (columns)
// (N <= 4: the rotation loops are unrolled, so the arrays are registers, not scratch memory)
template <int N>
__device__ void jacobi_eig(double (&a)[N][N], double (&v)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) v[i][j] = i == j ? 1.0 : 0.0;
#pragma clang loop unroll(disable)
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diag += a[i][i] * a[i][i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) off += a[i][j] * a[i][j];
        }
        if (off <= 1e-26 * diag || off == 0.0) break;     // off-diagonal norm below 1e-13 of the diagonal: converged in fp64
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// SVD of a 3x3 matrix through the eigen-decomposition of E^T E: singular values descending, U, V with u2 = u0 x u1, v2 = v0 x v1
__device__ void svd3(const double (&E)[3][3], double (&U)[3][3], double (&s)[3], double (&V)[3][3]) {
    double a[3][3], ev[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = E[0][i] * E[0][j] + E[1][i] * E[1][j] + E[2][i] * E[2][j];
    jacobi_eig<3>(a, ev);
    double lam[3] = {a[0][0], a[1][1], a[2][2]};
    // descending order by compare-and-swap of (eigenvalue, eigenvector column) pairs - static indices: everything stays in registers
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            const bool sw = lam[j] > lam[i];
            const double t = lam[i]; lam[i] = sw ? lam[j] : t; lam[j] = sw ? t : lam[j];
#pragma unroll
            for (int r = 0; r < 3; ++r) { const double u = ev[r][i]; ev[r][i] = sw ? ev[r][j] : u; ev[r][j] = sw ? u : ev[r][j]; }
        }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        s[c] = sqrt(fmax(lam[c], 0.0));
#pragma unroll
        for (int r = 0; r < 3; ++r) V[r][c] = ev[r][c];
#pragma unroll
        for (int r = 0; r < 3; ++r) U[r][c] = (E[r][0] * V[0][c] + E[r][1] * V[1][c] + E[r][2] * V[2][c]) / fmax(s[c], 1e-300);
    }
    s[2] = sqrt(fmax(lam[2], 0.0));
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
// The kernels take the weight from a table, like the published implementation of MAGSAC++ tabulates the incomplete gamma function:
// WLUT_N intervals in s = r / (k sigma_max) in [0, 1] (uniform in r, where w is smooth: w = 1 - O(r^3) at 0), linear interpolation,
// |table - closed form| < 1e-6.  erfc + exp in fp64 are ~400 instructions, and in the scoring kernel ONE lane of a wave inside the band of
// a bad model made the whole wave pay them: 0.1 ms per call at n = 3000.  The CPU twin reads the same table (oracle/pose_oracle.py).
constexpr int WLUT_N = 2048;
__device__ __forceinline__ double magsac_weight_lut(double r2, double inv_k2s2, const double* __restrict__ T) {
    const double s2 = r2 * inv_k2s2;                       // (r / (k sigma_max))^2
    if (!(s2 < 1.0)) return 0.0;
    const double u = sqrt(s2) * (double)WLUT_N;
    int j = (int)u;
    j = j < WLUT_N - 1 ? j : WLUT_N - 1;
    const double f = u - (double)j, a = T[j], b = T[j + 1];
    return a + f * (b - a);
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

// five-point sampler: a group of FP_L lanes per minimal sample (pose_fivept.h: the linear algebra of a solve is dealt to the lanes of
// the group, its matrices live in LDS), 64 / FP_L samples per wave; up to 10 models per sample, candidate c of sample h lives at slot
// 10 h + c.  History of this kernel at 1024 samples: one thread per sample with its arrays in scratch memory 1.84 ms, in LDS 0.65 ms,
// blocked inner loops 0.49 ms (73 % of it the QR iteration of the 10 x 10 action matrix), groups of 16 lanes: see profiles/r03.
// lanes per minimal sample (a power of two <= 64).  Round 3 (QR iteration: samples diverge by tens of sweeps): one wave per sample was 20 %
// faster than 16 lanes; with the root finder of round 4 the samples of a wave run nearly in step - the call takes the same 0.24-0.25 ms at
// 64, 32 and 16 lanes (profiles/r04/pose_lanes_per_sample_r4.log) and four samples per wave load the chip with 256 waves instead of 1024
// (the loops that run the pose step beside the matcher: +0-4 %)
#ifndef FP_L_N
#define FP_L_N 16
#endif
constexpr int FP_L = FP_L_N;
// hbase / need (round 5, adaptive termination): the launch covers samples hbase .. H - 1; with `need` a sample h >= *need is not drawn at all - its
// group marks its ten candidate slots invalid and leaves (a wave whose four samples are all beyond the bound exits at once)
__global__ __launch_bounds__(64) void pose_hypotheses5_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n,
                                                              int H, unsigned seed, double* __restrict__ Eh, int* __restrict__ valid,
                                                              int hbase, const int* __restrict__ need) {
    __shared__ fivept::Work work[64 / FP_L];
    const int g = threadIdx.x / FP_L, lane = threadIdx.x % FP_L;
    const int h = hbase + blockIdx.x * (64 / FP_L) + g;
    if (h >= H) return;                                     // whole groups leave together
    if (need != nullptr && h >= *need) {
        if (lane < 10) valid[(long)h * 10 + lane] = 0;
        return;
    }
    fivept::Work& w = work[g];
    int ids[5];
    bool ok = true;
    for (int k = 0; k < 5; ++k) {
        ids[k] = (int)(pose_rand(seed, (unsigned)h, (unsigned)k) % (unsigned)n);
        for (int j = 0; j < k; ++j) ok &= ids[j] != ids[k];
    }
    int nsol = 0;
    if (ok) {
        if (lane < 5) {
            int id = ids[0];
            for (int k = 1; k < 5; ++k) id = lane == k ? ids[k] : id;
            const double2 a = x0[id], b = x1[id];
            w.pts[lane][0] = a.x; w.pts[lane][1] = a.y; w.pts[lane][2] = b.x; w.pts[lane][3] = b.y;
        }
        fivept::group_fence<FP_L>();
#ifdef FP_PROFILE
        unsigned long long prof[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        nsol = fivept::five_point<FP_L>(Eh + (long)h * 90, w, lane, prof);
        if (lane == 0 && (h & 255) == 0)      // 100 MHz counter: x 10 ns
#ifdef FP_EIG_QR
            printf("fivept h=%d nsol=%d: null space %llu  cubics %llu  gauss-jordan %llu  eigenvalues %llu (hessenberg %llu, %llu QR sweeps, %llu bulge steps)  eigenvectors %llu  (x 10 ns)\n", h, nsol,
                   prof[1] - prof[0], prof[2] - prof[1], prof[3] - prof[2], prof[4] - prof[3], prof[6] - prof[3], prof[7], prof[8], prof[5] - prof[4]);
#else
            printf("fivept h=%d nsol=%d: null space %llu  cubics %llu  gauss-jordan %llu  eigenvalues %llu (hessenberg %llu, characteristic polynomial %llu, roots + polish %llu)  eigenvectors %llu  (x 10 ns)\n", h, nsol,
                   prof[1] - prof[0], prof[2] - prof[1], prof[3] - prof[2], prof[4] - prof[3], prof[6] - prof[3], prof[7] - prof[6], prof[4] - prof[7], prof[5] - prof[4]);
#endif
#else
        nsol = fivept::five_point<FP_L>(Eh + (long)h * 90, w, lane);
#endif
    }
    if (lane < 10) valid[(long)h * 10 + lane] = lane < nsol ? 1 : 0;
}

// Quality of every candidate model: inlier count (magsac == 0) or sigma-marginalised quality floor(4096 sum_i w(r_i)) (magsac != 0).
// One workgroup per SCORE_C consecutive candidate slots; SCORE_C = 1 (measured at n = 3000: 43 us; 5 candidates per workgroup share
// the point loads but are 79 us, 16 in a loop 164 us - the kernel lives on the parallelism of ~10 k independent workgroups, not on
// memory traffic).  Per candidate the summation order is fixed: thread-strided partial sums, wave shuffle tree, (w0 + w1) + (w2 + w3).
#ifndef SCORE_C_N
#define SCORE_C_N 1
#endif
constexpr int SCORE_C = SCORE_C_N;
__global__ __launch_bounds__(256) void pose_score_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n, int ncand,
                                                         const double* __restrict__ Eh, const int* __restrict__ valid, double thr2,
                                                         const double* __restrict__ wlut, int* __restrict__ counts, int cbase) {
    __shared__ double red[SCORE_C][4];
    const bool magsac = wlut != nullptr;
    const double inv_k2s2 = 1.0 / (MAGSAC_K * MAGSAC_K * thr2);
    const int h0 = cbase + blockIdx.x * SCORE_C;
    double E[SCORE_C][9], c[SCORE_C];
    bool ok[SCORE_C];
    bool any = false;
#pragma unroll
    for (int k = 0; k < SCORE_C; ++k) {
        ok[k] = h0 + k < ncand && valid[h0 + k < ncand ? h0 + k : 0];
        any |= ok[k];
        c[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) E[k][i] = ok[k] ? Eh[(long)(h0 + k) * 9 + i] : 0.0;
    }
    if (any) {                                             // uniform
        // most points of most candidates are far outside: those are recognised without the division of the Sampson distance
        // (num^2 >= limit x den with a margin far above the rounding of either side); everything else takes the exact path
        const double limit = (magsac ? MAGSAC_K * MAGSAC_K * thr2 : thr2) * (1.0 + 1e-9);
        for (int i = threadIdx.x; i < n; i += 256) {
            const double2 p0 = x0[i], p1 = x1[i];
#pragma unroll
            for (int k = 0; k < SCORE_C; ++k) {
                if (!ok[k]) continue;                      // uniform
                const double* Ek = E[k];
                const double a0 = Ek[0] * p0.x + Ek[1] * p0.y + Ek[2], a1 = Ek[3] * p0.x + Ek[4] * p0.y + Ek[5], a2 = Ek[6] * p0.x + Ek[7] * p0.y + Ek[8];
                const double b0 = Ek[0] * p1.x + Ek[3] * p1.y + Ek[6], b1 = Ek[1] * p1.x + Ek[4] * p1.y + Ek[7];
                const double num = p1.x * a0 + p1.y * a1 + a2;
                const double den = fmax(a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1, 1e-30);
                if (num * num >= limit * den) continue;
                const double r2 = num * num / den;
                c[k] += magsac ? magsac_weight_lut(r2, inv_k2s2, wlut) : (r2 < thr2 ? 1.0 : 0.0);
            }
        }
#pragma unroll
        for (int k = 0; k < SCORE_C; ++k) {
            for (int o = 32; o > 0; o >>= 1) c[k] += __shfl_xor(c[k], o);
            if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = c[k];
        }
        __syncthreads();
    }
    if (threadIdx.x < SCORE_C && h0 + (int)threadIdx.x < ncand) {
        const int k = threadIdx.x;
        bool okk = false;
#pragma unroll
        for (int j = 0; j < SCORE_C; ++j) okk = j == k ? ok[j] : okk;
        const double q = any ? (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]) : 0.0;
        counts[h0 + k] = okk ? (magsac ? (int)floor(q * QUALITY_SCALE) : (int)q) : -1;
    }
}

// Adaptive termination (round 5; the rule of RANSAC / USAC, which the reference reaches through cv2.findEssentialMat(..., prob = 0.99999,
// method = USAC_MAGSAC), eval/pose_estimation.py:96-105 - OpenCV's own implementation stays unpinned): after the first H1 samples the best
// support so far gives the inlier ratio w, and k samples contain an all-inlier one with probability 1 - (1 - w^5)^k; the smallest k
// with (1 - w^5)^k <= 1 - 0.99999 is all that is drawn (at least H1, at most Hmax).  The bound is found by repeated multiplication in
// IEEE doubles - no log(), so that the CPU twin (oracle/pose_oracle.py) computes the same integer bit for bit.  MAGSAC scoring: the
// quality floor(4096 sum w_i) / 4096 is a SOFT inlier count (every weight <= 1): it under-states w, the bound errs on the side of more samples.
constexpr int POSE_H1 = 128;
__global__ __launch_bounds__(256) void pose_need_kernel(const int* __restrict__ counts, int ncand1, int n, int magsac, int Hmax, int* __restrict__ need) {
    __shared__ int sm[4];
    int best = -1;
    for (int h = threadIdx.x; h < ncand1; h += 256) best = max(best, counts[h]);
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
        int k = Hmax;
        if (best > 0) {
            double w = magsac ? (double)best / (QUALITY_SCALE * (double)n) : (double)best / (double)n;
            if (w > 1.0) w = 1.0;
            const double w5 = w * w * w * w * w;
            const double pfail = 1.0 - w5;
            double q = 1.0;
            k = 0;
            while (k < Hmax) { q = q * pfail; ++k; if (q <= 1e-5) break; }
        }
        *need = k < POSE_H1 ? POSE_H1 : k;
    }
}

// ---- refinement of the best hypothesis ----------------------------------------------------------------------------------------------
constexpr int CT = 512;            // threads of the consensus workgroup
constexpr int CPT = 8;             // correspondences a thread keeps in registers (coordinates + current weight): n <= 4096; beyond that recomputed

// value of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ double wave_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// sums of K per-thread values over the workgroup (CT threads = 8 waves): wave reduction, then the 8 partials in a fixed order.
// The totals land in red[0 .. K); the caller reads them after the trailing barrier.
template <int K>
__device__ void block_sums(double (&v)[K], double* part, double* red) {
#pragma unroll
    for (int k = 0; k < K; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();                                   // the previous user of part / red is done
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) part[(threadIdx.x >> 6) * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double t = 0;
        for (int w = 0; w < CT / 64; ++w) t += part[w * K + threadIdx.x];
        red[threadIdx.x] = t;
    }
    __syncthreads();
}

// Eigenvector of the smallest eigenvalue of the 9 x 9 normal matrix M (LDS, row-major, symmetric positive semi-definite), by ONE wave:
// LDL^T factorisation of M + delta I and inverse iteration from v0 - the conditioned current model, already close to the answer, so
// every iteration multiplies the error by lambda_1 / lambda_2 of a matrix whose smallest eigenvalue is the (tiny) fit residual.
// Lane i owns row i of L and component i of the vectors; pivots and solution components travel by v_readlane.  (A cyclic Jacobi sweep
// of the same matrix is 36 dependent rotations of ~1000 cycles each: 0.1 ms per fit, half of this kernel before.)
__device__ void smallest_eigvec9_wave(const double* M, const double* v0, double* vout, int lane) {
    double Lr[9];                                        // row `lane` of the factor (columns < lane), D on the diagonal
    const int li = lane < 9 ? lane : 8;
    double tr = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) { Lr[j] = M[li * 9 + j]; tr += M[j * 9 + j]; }
    const double delta = 1e-15 * tr + 1e-300;
#pragma unroll
    for (int j = 0; j < 9; ++j) Lr[j] += j == li ? delta : 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double dk = wave_bcast(Lr[k], k);          // D_k (row k, column k)
        const double lik = Lr[k] / dk;                   // l_ik for the rows below k
#pragma unroll
        for (int j = k + 1; j < 9; ++j) {
            const double ajk = wave_bcast(Lr[k], j);     // a_jk = l_jk D_k, still unscaled in row j
            Lr[j] -= li > k ? lik * ajk : 0.0;           // only the lower triangle (j <= i) is used later
        }
        Lr[k] = li > k ? lik : Lr[k];
    }
    double Lt[9];                                        // column `lane` of L below the diagonal: Lt[k] = l_k,lane
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) { const double x = wave_bcast(Lr[j], k); t = j == li ? x : t; }
        Lt[k] = t;
    }
    double v = v0[li];
    {
        double nn = lane < 9 ? v * v : 0.0;
        for (int o = 8; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 16);
        v /= sqrt(nn);
    }
    double dii = Lr[0];                                  // D_i sits on the diagonal of row i
#pragma unroll
    for (int j = 1; j < 9; ++j) dii = li == j ? Lr[j] : dii;
#pragma clang loop unroll(disable)
    for (int it = 0; it < 24; ++it) {
        double b = v;
#pragma unroll
        for (int k = 0; k < 9; ++k) {                    // L y = b
            const double yk = wave_bcast(b, k);
            b -= li > k ? Lr[k] * yk : 0.0;
        }
        b /= dii;                                        // D z = y
#pragma unroll
        for (int k = 8; k >= 0; --k) {                   // L^T x = z
            const double xk = wave_bcast(b, k);
            b -= li < k ? Lt[k] * xk : 0.0;
        }
        double nn = lane < 9 ? b * b : 0.0, dot = lane < 9 ? b * v : 0.0;
        for (int o = 8; o > 0; o >>= 1) { nn += __shfl_xor(nn, o, 16); dot += __shfl_xor(dot, o, 16); }
        nn = wave_bcast(nn, 0); dot = wave_bcast(dot, 0);
        const double x = b / (dot >= 0.0 ? sqrt(nn) : -sqrt(nn));
        double diff = lane < 9 ? fabs(x - v) : 0.0;
        for (int o = 8; o > 0; o >>= 1) diff = fmax(diff, __shfl_xor(diff, o, 16));
        diff = wave_bcast(diff, 0);
        v = x;
        if (diff <= 2e-15) break;                        // wave-uniform
    }
    if (lane < 9) vout[lane] = v;
}

// one workgroup: first best hypothesis -> (weighted) least-squares refits kept while not worse -> decomposition of E
__global__ __launch_bounds__(CT) void pose_consensus_kernel(const double2* __restrict__ x0, const double2* __restrict__ x1, int n, int H,
                                                           const double* __restrict__ Eh, const int* __restrict__ counts, double thr2, const double* __restrict__ wlut, int nsample,
                                                           unsigned char* __restrict__ inl, double* __restrict__ out, int* __restrict__ good) {
    // out: [0..8] E, [9..17] R, [18..20] t, [21] inliers of E, [22] cheirality inliers, [23] ok flag, [24..32] R1, [33..41] R2, [42..44] t
    __shared__ double part[(CT / 64) * 45], red[48];
    __shared__ double Es[9], Et[9], JA[81], Fv[9], F0[9];
    __shared__ int s_best, s_cnt, s_ok;
    const int tid = threadIdx.x;
    const bool magsac = wlut != nullptr;
    const double inv_k2s2 = 1.0 / (MAGSAC_K * MAGSAC_K * thr2);
    if (tid < 4) good[tid] = 0;                                   // the vote counters of the cheirality kernel that follows
    {   // first best hypothesis (largest count, lowest index): strided scan + wave / workgroup reduction
        int best = -1, bi = 0x7fffffff;
        for (int h = tid; h < H; h += CT) { const int c = counts[h]; if (c > best) { best = c; bi = h; } }
        for (int o = 32; o > 0; o >>= 1) {
            const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        int* smi = reinterpret_cast<int*>(part);
        if ((tid & 63) == 0) { smi[2 * (tid >> 6)] = best; smi[2 * (tid >> 6) + 1] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < CT / 64; ++w)
                if (smi[2 * w] > best || (smi[2 * w] == best && smi[2 * w + 1] < bi)) { best = smi[2 * w]; bi = smi[2 * w + 1]; }
            s_best = best >= 0 ? bi : -1; s_cnt = best;
            if (best >= 0) for (int i = 0; i < 9; ++i) Es[i] = Eh[(long)bi * 9 + i];
        }
        __syncthreads();
    }
    // nsample = size of the minimal sample (5 or 8): a model has to explain at least that many matches (half of it in weight units)
    if (s_best < 0 || s_cnt < (magsac ? (int)(nsample * QUALITY_SCALE / 2) : nsample)) { if (tid == 0) out[23] = 0.0; return; }
    // per-point weight of a model: 0 / 1 membership of the consensus set, or the sigma-marginalised weight (IRLS)
    auto weight_of = [&](const double* E, double ax, double ay, double bx, double by) {
        const double r2 = sampson_sq(E, ax, ay, bx, by);
        return magsac ? magsac_weight_lut(r2, inv_k2s2, wlut) : (r2 < thr2 ? 1.0 : 0.0);
    };
    // the first CPT points of a thread stay on chip - their coordinates in LDS (128 KB: this is the only workgroup of the launch; in
    // registers they made the kernel spill), their weight under the current model in registers; the rest (n > 4096) is recomputed
    extern __shared__ __attribute__((aligned(16))) double2 cpts[];
    double2* const sa = cpts + tid;                              // point u of this thread: sa[u * CT], sb[u * CT] (16-byte stride over the lanes)
    double2* const sb = cpts + CPT * CT + tid;
    double pw[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int i = tid + u * CT;
        const bool in = i < n;
        const double2 a = x0[in ? i : 0], b = x1[in ? i : 0];
        sa[u * CT] = a; sb[u * CT] = b;                          // (read back by this thread only: no barrier)
        pw[u] = in ? weight_of(Es, a.x, a.y, b.x, b.y) : 0.0;
        if (in) inl[i] = sampson_sq(Es, a.x, a.y, b.x, b.y) < thr2;
    }
    for (int i = tid + CPT * CT; i < n; i += CT) inl[i] = sampson_sq(Es, x0[i].x, x0[i].y, x1[i].x, x1[i].y) < thr2;
    // f(ax, ay, bx, by, w) over all points with their weight under Es
    auto for_points = [&](auto&& f) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) if (pw[u] > 0) { const double2 a = sa[u * CT], b = sb[u * CT]; f(a.x, a.y, b.x, b.y, pw[u]); }
        for (int i = tid + CPT * CT; i < n; i += CT) {
            const double w = weight_of(Es, x0[i].x, x0[i].y, x1[i].x, x1[i].y);
            if (w > 0) f(x0[i].x, x0[i].y, x1[i].x, x1[i].y, w);
        }
    };
    for (int round = 0; round < (n >= 8 ? 3 : 0); ++round) {      // the linear refit needs 8 correspondences
        // conditioning of the (weighted) consensus set
        double m5[5] = {0, 0, 0, 0, 0};
        for_points([&](double ax, double ay, double bx, double by, double w) { m5[0] += w; m5[1] += w * ax; m5[2] += w * ay; m5[3] += w * bx; m5[4] += w * by; });
        block_sums<5>(m5, part, red);
        const double cnt = red[0], cx0 = red[1] / cnt, cy0 = red[2] / cnt, cx1 = red[3] / cnt, cy1 = red[4] / cnt;
        double d2[2] = {0, 0};
        for_points([&](double ax, double ay, double bx, double by, double w) {
            d2[0] += w * sqrt((ax - cx0) * (ax - cx0) + (ay - cy0) * (ay - cy0));
            d2[1] += w * sqrt((bx - cx1) * (bx - cx1) + (by - cy1) * (by - cy1));
        });
        block_sums<2>(d2, part, red);
        const double s0 = 1.4142135623730951 / fmax(red[0] / cnt, 1e-12), s1 = 1.4142135623730951 / fmax(red[1] / cnt, 1e-12);
        // the normal matrix sum_i w_i r_i r_i^T of the rows r = (bx, by, 1) (x) (ax, ay, 1) is sum_i w_i (b b^T) (x) (a a^T): 6 x 6 = 36
        // distinct sums instead of the 45 of a general symmetric 9 x 9 matrix (fewer live registers: the kernel used to spill)
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        for_points([&](double ax0, double ay0, double bx0, double by0, double w) {
            const double ax = (ax0 - cx0) * s0, ay = (ay0 - cy0) * s0, bx = (bx0 - cx1) * s1, by = (by0 - cy1) * s1;
            const double qa[6] = {ax * ax, ax * ay, ax, ay * ay, ay, 1.0};                      // a a^T: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
            const double qb[6] = {w * bx * bx, w * bx * by, w * bx, w * by * by, w * by, w};    // w b b^T, same order
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int v = 0; v < 6; ++v) acc[u * 6 + v] += qb[u] * qa[v];
        });
        block_sums<36>(acc, part, red);
        if (tid < 81) {                                            // entry (3 i + k, 3 j + l) = (b b^T)_ij (a a^T)_kl
            const int r = tid / 9, c = tid - 9 * r;
            const int i = r / 3, k = r - 3 * i, j = c / 3, l = c - 3 * j;
            const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
            JA[tid] = red[sym[i][j] * 6 + sym[k][l]];
        }
        if (tid == 64) {                                           // the current model in conditioned coordinates: F0 = T1^-T Es T0^-1
            // T^-1 = [[1/s, 0, cx], [0, 1/s, cy], [0, 0, 1]]
            double G[3][3];
            for (int r = 0; r < 3; ++r) {
                G[r][0] = Es[r * 3] / s0; G[r][1] = Es[r * 3 + 1] / s0;
                G[r][2] = Es[r * 3] * cx0 + Es[r * 3 + 1] * cy0 + Es[r * 3 + 2];
            }
            for (int c = 0; c < 3; ++c) {
                F0[c] = G[0][c] / s1; F0[3 + c] = G[1][c] / s1;
                F0[6 + c] = cx1 * G[0][c] + cy1 * G[1][c] + G[2][c];
            }
        }
        __syncthreads();
        if (tid < 64) smallest_eigvec9_wave(JA, F0, Fv, tid);
        __syncthreads();
        if (tid == 0) {
            double F[3][3];
            for (int i = 0; i < 9; ++i) F[i / 3][i % 3] = Fv[i];
            const double T0[3] = {s0, cx0, cy0}, T1[3] = {s1, cx1, cy1};
            double E2[3][3];
            s_ok = essential_from_F(F, T0, T1, E2) ? 1 : 0;
            if (s_ok) for (int i = 0; i < 9; ++i) Et[i] = E2[i / 3][i % 3];
        }
        __syncthreads();
        if (!s_ok) break;
        double c2[1] = {0};
#pragma unroll
        for (int u = 0; u < CPT; ++u) { const double2 a = sa[u * CT], b = sb[u * CT]; c2[0] += tid + u * CT < n ? weight_of(Et, a.x, a.y, b.x, b.y) : 0.0; }
        for (int i = tid + CPT * CT; i < n; i += CT) c2[0] += weight_of(Et, x0[i].x, x0[i].y, x1[i].x, x1[i].y);
        block_sums<1>(c2, part, red);
        const int cnt2 = magsac ? (int)floor(red[0] * QUALITY_SCALE) : (int)red[0];
        if (cnt2 < s_cnt) break;                                   // uniform: kept only while not worse
        const bool same = cnt2 == s_cnt;
        __syncthreads();
        if (tid == 0) { s_cnt = cnt2; for (int i = 0; i < 9; ++i) Es[i] = Et[i]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int i = tid + u * CT;
            const double2 a = sa[u * CT], b = sb[u * CT];
            pw[u] = i < n ? weight_of(Es, a.x, a.y, b.x, b.y) : 0.0;                             // (recomputed rather than held)
            if (i < n) inl[i] = sampson_sq(Es, a.x, a.y, b.x, b.y) < thr2;
        }
        for (int i = tid + CPT * CT; i < n; i += CT) inl[i] = sampson_sq(Es, x0[i].x, x0[i].y, x1[i].x, x1[i].y) < thr2;
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
    // one thread per (point, candidate): the four lanes of a point sit next to each other in the wave
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 2, k = t & 3;
    const bool live = out[23] != 0.0 && i < n && inl[i < n ? i : 0];
    unsigned bb = 0;
    if (live) {
        const double* R = out + 24 + (k & 1) * 9;
        const double sg = k >= 2 ? -1.0 : 1.0;
        const double P[3][4] = {{R[0], R[1], R[2], sg * out[42]}, {R[3], R[4], R[5], sg * out[43]}, {R[6], R[7], R[8], sg * out[44]}};
        const double2 a = x0[i], b = x1[i];
        // DLT rows (cv2.triangulatePoints): x P0[2] - P0[0], y P0[2] - P0[1] with P0 = [I | 0], and the same for P
        const double A[4][4] = {{-1, 0, a.x, 0}, {0, -1, a.y, 0},
                                {b.x * P[2][0] - P[0][0], b.x * P[2][1] - P[0][1], b.x * P[2][2] - P[0][2], b.x * P[2][3] - P[0][3]},
                                {b.y * P[2][0] - P[1][0], b.y * P[2][1] - P[1][1], b.y * P[2][2] - P[1][2], b.y * P[2][3] - P[1][3]}};
        double ata[4][4], ev[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) ata[r][c] = A[0][r] * A[0][c] + A[1][r] * A[1][c] + A[2][r] * A[2][c] + A[3][r] * A[3][c];
        jacobi_eig<4>(ata, ev);
        double lmin = ata[0][0];
        double Q[4] = {ev[0][0], ev[1][0], ev[2][0], ev[3][0]};
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const bool lt = ata[c][c] < lmin;
            lmin = lt ? ata[c][c] : lmin;
#pragma unroll
            for (int r = 0; r < 4; ++r) Q[r] = lt ? ev[r][c] : Q[r];
        }
        bool ok = Q[2] * Q[3] > 0;
        const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
        ok = ok && Z < dist_thresh;
        const double zc = P[2][0] * X + P[2][1] * Y + P[2][2] * Z + P[2][3];
        ok = ok && zc > 0 && zc < dist_thresh;
        if (ok) { bb = 1u << k; atomicAdd(&cnt[k], 1); }
    }
    bb |= __shfl_xor(bb, 1);
    bb |= __shfl_xor(bb, 2);
    if (live && k == 0) bits[i] = (unsigned char)bb;
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
    if (blockIdx.x == 0 && threadIdx.x == 0) out[45] = (double)good[4];        // samples drawn (adaptive termination): travels with the results
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

// pixel coordinates (float, in the host's pinned staging buffer, read over the bus by this kernel) -> normalised camera coordinates:
// (x - cx) / fx, (y - cy) / fy per camera, in fp64 like the host loop it replaces.  A hipMemcpyAsync of more than a few KB takes the
// runtime's staged path here: ~55-70 us before the first kernel starts; this is 3-5 us.
__global__ __launch_bounds__(256) void pose_upload_kernel(const float* __restrict__ k0, const float* __restrict__ k1, int n, double cx0, double fx0,
                                                          double cy0, double fy0, double cx1, double fx1, double cy1, double fy1,
                                                          double2* __restrict__ x0, double2* __restrict__ x1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 a = reinterpret_cast<const float2*>(k0)[i], b = reinterpret_cast<const float2*>(k1)[i];
    x0[i] = double2{((double)a.x - cx0) / fx0, ((double)a.y - cy0) / fy0};
    x1[i] = double2{((double)b.x - cx1) / fx1, ((double)b.y - cy1) / fy1};
}

// per-thread workspace.  Device: x = [x0 | x1], res = [out 48 doubles | refmask | inl] (one read-back); host: pinned
// staging for both, so neither copy goes through the runtime's pageable-memory path (three staged uploads cost ~0.15 ms per call)
struct PoseWs {
    int device = -1;
    size_t cap_n = 0, cap_h = 0;
    double2* x = nullptr;
    unsigned char* res = nullptr;
    double* Eh = nullptr;
    int *valid = nullptr, *counts = nullptr, *good = nullptr;
    unsigned char* bits = nullptr;
    double* wlut = nullptr;            // MAGSAC++ weight table (WLUT_N + 1 doubles)
    unsigned char* pin = nullptr;      // hipHostMalloc: the float keypoints on the way in (16 n bytes), the results on the way out (384 + 2 n bytes)
    unsigned char* pin_dev = nullptr;  // the same buffer as the device sees it
};

}  // namespace

static std::atomic<long> g_pose_calls{0}, g_pose_samples{0};
// process-wide counters of the pose step: calls and minimal samples drawn (adaptive termination: fewer than calls x iterations)
extern "C" void imp_pose_stats(long* calls, long* samples, int reset) {
    if (calls) *calls = g_pose_calls.load();
    if (samples) *samples = g_pose_samples.load();
    if (reset) { g_pose_calls.store(0); g_pose_samples.store(0); }
}

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
        for (void* p : {(void*)ws.x, (void*)ws.res, (void*)ws.Eh, (void*)ws.valid, (void*)ws.counts, (void*)ws.good, (void*)ws.bits, (void*)ws.wlut})
            if (p) (void)hipFree(p);
        if (ws.pin) (void)hipHostFree(ws.pin);
        ws = PoseWs();
        const size_t cn = (size_t)n < 4096 ? 4096 : (size_t)n, ch = (size_t)iterations < 2048 ? 2048 : (size_t)iterations;
        if (hipMalloc(&ws.x, 2 * cn * sizeof(double2)) != hipSuccess || hipMalloc(&ws.res, 48 * sizeof(double) + 2 * cn) != hipSuccess ||
            hipMalloc(&ws.Eh, ch * 10 * 9 * sizeof(double)) != hipSuccess || hipMalloc(&ws.good, 8 * sizeof(int)) != hipSuccess ||
            hipMalloc(&ws.bits, cn) != hipSuccess || hipMalloc(&ws.valid, ch * 10 * sizeof(int)) != hipSuccess ||
            hipMalloc(&ws.counts, ch * 10 * sizeof(int)) != hipSuccess || hipHostMalloc(&ws.pin, 2 * cn * sizeof(double2)) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&ws.pin_dev), ws.pin, 0) != hipSuccess ||
            hipMalloc(&ws.wlut, (WLUT_N + 1) * sizeof(double)) != hipSuccess)
            return IMP_E_NOMEM;
        {   // w at s = j / WLUT_N, i.e. r = s k sigma_max (sigma_max = 1); the last entry is the cut-off: exactly 0
            double* T = reinterpret_cast<double*>(ws.pin);
            for (int j = 0; j <= WLUT_N; ++j) {
                const double sj = (double)j / (double)WLUT_N;
                T[j] = j == WLUT_N ? 0.0 : magsac_weight(sj * sj * MAGSAC_K * MAGSAC_K, 1.0);
            }
            if (hipMemcpy(ws.wlut, T, (WLUT_N + 1) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return IMP_E_HIP;
        }
        ws.device = device; ws.cap_n = cn; ws.cap_h = ch;
    }
    double2* const x0 = ws.x;
    double2* const x1 = ws.x + n;
    double* const dout = reinterpret_cast<double*>(ws.res);
    unsigned char* const refmask = ws.res + 48 * sizeof(double);
    unsigned char* const inl = refmask + n;
    memcpy(ws.pin, kpts0, (size_t)n * 2 * sizeof(float));
    memcpy(ws.pin + (size_t)n * 2 * sizeof(float), kpts1, (size_t)n * 2 * sizeof(float));
    hipLaunchKernelGGL(pose_upload_kernel, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float*>(ws.pin_dev),
                       reinterpret_cast<const float*>(ws.pin_dev) + (size_t)n * 2, n, K0[2], K0[0], K0[5], K0[4], K1[2], K1[0], K1[5], K1[4], x0, x1);
    const double thr = norm_thresh / ((K0[0] + K0[4] + K1[0] + K1[4]) / 4.0);
    const int ncand = eight ? iterations : iterations * 10;
    constexpr int SPW = 64 / FP_L;                         // samples per workgroup of the five-point kernel
    const bool adaptive = (flags & 4) != 0 && !eight && iterations > POSE_H1;
    if (eight) {
        hipLaunchKernelGGL(pose_hypotheses_kernel, dim3((iterations + 63) / 64), dim3(64), 0, st, x0, x1, n, iterations, seed, ws.Eh, ws.valid);
        hipLaunchKernelGGL(pose_score_kernel, dim3((ncand + SCORE_C - 1) / SCORE_C), dim3(256), 0, st, x0, x1, n, ncand, ws.Eh, ws.valid, thr * thr, (flags & 1) ? ws.wlut : nullptr, ws.counts, 0);
    } else if (!adaptive) {
        hipLaunchKernelGGL(pose_hypotheses5_kernel, dim3((iterations + SPW - 1) / SPW), dim3(64), 0, st, x0, x1, n, iterations, seed, ws.Eh, ws.valid, 0, (const int*)nullptr);
        hipLaunchKernelGGL(pose_score_kernel, dim3((ncand + SCORE_C - 1) / SCORE_C), dim3(256), 0, st, x0, x1, n, ncand, ws.Eh, ws.valid, thr * thr, (flags & 1) ? ws.wlut : nullptr, ws.counts, 0);
    } else {
        // IMP_POSE_ADAPTIVE: the first POSE_H1 samples and their scores, the bound from the best support (on the device: nothing here waits), then
        // the remaining samples in launches whose groups / workgroups beyond the bound leave at once - at 70 % inliers 63 samples suffice, the
        // fixed budget drew 1024: an eighth of the waves, and of the compute units they keep from the matcher's LDS-filling kernels
        const int nc1 = POSE_H1 * 10;
        hipLaunchKernelGGL(pose_hypotheses5_kernel, dim3(POSE_H1 / SPW), dim3(64), 0, st, x0, x1, n, POSE_H1, seed, ws.Eh, ws.valid, 0, (const int*)nullptr);
        hipLaunchKernelGGL(pose_score_kernel, dim3((nc1 + SCORE_C - 1) / SCORE_C), dim3(256), 0, st, x0, x1, n, nc1, ws.Eh, ws.valid, thr * thr, (flags & 1) ? ws.wlut : nullptr, ws.counts, 0);
        hipLaunchKernelGGL(pose_need_kernel, dim3(1), dim3(256), 0, st, ws.counts, nc1, n, (flags & 1) ? 1 : 0, iterations, ws.good + 4);
        hipLaunchKernelGGL(pose_hypotheses5_kernel, dim3((iterations - POSE_H1 + SPW - 1) / SPW), dim3(64), 0, st, x0, x1, n, iterations, seed, ws.Eh, ws.valid, POSE_H1, (const int*)(ws.good + 4));
        hipLaunchKernelGGL(pose_score_kernel, dim3((ncand - nc1 + SCORE_C - 1) / SCORE_C), dim3(256), 0, st, x0, x1, n, ncand, ws.Eh, ws.valid, thr * thr, (flags & 1) ? ws.wlut : nullptr, ws.counts, nc1);
    }
    // the cheirality step of the reference normalises with K = (K0 + K1) / 2 (eval/pose_estimation.py:29-33): with K0 == K1 (every
    // caller in the repo) these are the coordinates above; a caller with two different cameras gets per-camera normalisation
    constexpr size_t consensus_lds = 2 * (size_t)CPT * CT * sizeof(double2);
    if (imp_grant_dynamic_lds(reinterpret_cast<const void*>(&pose_consensus_kernel), consensus_lds) != hipSuccess) return IMP_E_HIP;
    hipLaunchKernelGGL(pose_consensus_kernel, dim3(1), dim3(CT), consensus_lds, st, x0, x1, n, ncand, ws.Eh, ws.counts, thr * thr, (flags & 1) ? ws.wlut : nullptr, eight ? 8 : 5, inl, dout, ws.good);
    hipLaunchKernelGGL(pose_cheirality_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, st, x0, x1, n, dout, inl, 1000.0, ws.bits, ws.good);
    hipLaunchKernelGGL(pose_vote_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, ws.good, ws.bits, inl, refmask, dout);
    // one read-back: the 24 result doubles, the reference-semantics mask and the geometric mask
    const size_t res_bytes = 48 * sizeof(double) + 2 * (size_t)n;
    if (hipMemcpyAsync(ws.pin, ws.res, res_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return IMP_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return IMP_E_HIP;
    g_pose_calls.fetch_add(1, std::memory_order_relaxed);
    g_pose_samples.fetch_add(adaptive ? (long)reinterpret_cast<const double*>(ws.pin)[45] : (long)iterations, std::memory_order_relaxed);
    if (hipGetLastError() != hipSuccess) return IMP_E_HIP;
    const double* out = reinterpret_cast<const double*>(ws.pin);
    memcpy(mask, ws.pin + 48 * sizeof(double), n);
    if (consensus) memcpy(consensus, ws.pin + 48 * sizeof(double) + n, n);
    if (out[23] == 0.0) return 1;
    for (int i = 0; i < 9; ++i) { E[i] = out[i]; R[i] = out[9 + i]; }
    for (int i = 0; i < 3; ++i) t[i] = out[18 + i];
    *n_inliers = (int)out[22];
    return IMP_OK;
}
