// Five-point relative pose (minimal solver of cv2.findEssentialMat, eval/pose_estimation.py:96-105), restated from the publications -
// D. Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004, in the formulation of H. Stewenius, C. Engels,
// D. Nister, "Recent developments on direct relative orientation", ISPRS J. 2006:
//   E = x X + y Y + z Z + W in the 4-dimensional null space of the five epipolar constraints; det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0
//   are ten cubics in (x, y, z); eliminating the ten degree-3 monomials leaves the multiplication-by-x map on the quotient-ring basis
//   [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1] as a 10 x 10 matrix whose real eigenpairs are the (up to ten) solutions.
// fp64.  CPU twin: oracle/pose_oracle.py five_point (same elimination order, same candidate order).
//
// One solve is executed by a GROUP of L lanes of a wave over matrices in LDS (struct Work); L = 1 is plain sequential C++ and is what the
// host unit test of the algebra builds (tests/test_pose.py compiles tools/probe/fivept_host.cpp with g++).  Why a group, measured on
// gfx950 with 1024 samples per call:
//   * one thread per sample, run-time-indexed arrays in scratch memory: an L2 round trip per access, ~25 k of them: 1.84 ms per call;
//   * the same in LDS, element by element: 0.65 ms; inner loops as fixed-width register blocks: 0.49 ms - of which the eigenvalues of the
//     10 x 10 action matrix (Hessenberg + ~30 double-shift QR sweeps) are 0.33 ms: one lane issues every instruction of every sweep;
//   * everything unrolled into registers: 112 KB of straight-line code per wave, the instruction cache becomes the bottleneck (0.49 ms).
// So the rows / columns of every elimination step and of every QR sweep are dealt to the lanes of the group (`for (j = j0 + lane; j <= j1;
// j += L)`), the scalar bookkeeping between them is computed redundantly by all lanes from LDS (same values, same control flow for the
// whole group), pivot searches are group reductions, and the (up to ten) eigenvector systems are solved one per lane.  Within a group
// the LDS operations of consecutive instructions execute in program order (one wave), so the only synchronisation is a compiler fence.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define FP_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define FP_HD inline
#endif
#if defined(__clang__)
#define FP_UNROLL _Pragma("unroll")
#define FP_LOOP _Pragma("clang loop unroll(disable)")
#else
#define FP_UNROLL
#define FP_LOOP
#endif

#if defined(FP_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define FP_STAMP(k) prof[k] = wall_clock64()
#define FP_COUNT(k) prof[k] += 1
#else
#define FP_STAMP(k)
#define FP_COUNT(k)
#endif

namespace fivept {

// scratch of the root finder (aliases Work::G, which is only needed once the eigenvalues are known)
struct RootScratch {
    double P[11][12];          // P[k]: characteristic polynomial of the leading k x k block of the Hessenberg matrix (lambda^0 .. lambda^k, zero above)
    double Q[10][12];          // Q[m] = p^(m) / m!: Q[m][j] = C(j + m, m) c[j + m] (zero above degree 10 - m)
    double rt[2][12];          // the real roots of two consecutive derivative levels, ascending
    double cand[12];           // level being processed: the root of interval i ...
    int found[12];             // ... if it holds one
};

// Scratch of one solve (LDS on the GPU): 8240 bytes
struct Work {
    double pts[5][4];          // the sample: x0, y0, x1, y1 (normalised coordinates)
    double Q[5][9];            // epipolar constraints -> reduced row echelon form
    double B[4][9];            // null-space basis X, Y, Z, W: entry (i, j) of E is the linear polynomial B[0..3][3 i + j] in [x, y, z, 1]
    double minors[3][10];      // 2 x 2 minors of rows 1, 2 of E (quadratic polynomials)
    double EEt[9][10];         // E E^T
    double tr[10];             // trace(E E^T)
    double A[10][20];          // the ten cubics [degree-3 monomials | quotient basis] -> A1^-1 A2
    double H[11][11];          // action matrix, 1-based (EISPACK)
    double wr[11], wi[11];     // eigenvalues; wr is reused for the sorted real ones
    union {
        double G[10][6][6];    // eigenvector system of each real eigenvalue
        RootScratch rs;        // (before that: the root finder's tables)
    };
    double Es[10][9];          // candidate essential matrices
    int ok[10];
    int pad[2];
};

// ---- group primitives (L lanes of one wave; L = 1: nothing to do) ----
template <int L> FP_HD void group_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (L > 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#endif
}
// arg max over the group: the largest v wins, the smallest key among equal v (keys are unique per candidate)
template <int L> FP_HD void group_argmax(double& v, int& key) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (L > 1) {
        FP_UNROLL for (int o = L / 2; o > 0; o >>= 1) {
            const double ov = __shfl_xor(v, o, L);
            const int ok = __shfl_xor(key, o, L);
            const bool take = ov > v || (ov == v && ok < key);
            v = take ? ov : v; key = take ? ok : key;
        }
    }
#endif
}

// the value lane `src` of the group holds (src is the same on every lane of the group)
template <int L> FP_HD double group_bcast(double v, int src) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (L == 64) {                                         // the group is the wave: src is wave-uniform, v_readlane needs no LDS round trip
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    }
    if (L > 1) return __shfl(v, src, L);
#endif
    return v;
}

// slot tables of the polynomial products (monomial orders: linear [x, y, z, 1]; quadratic [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1];
// cubic [x^3, x^2 y, x^2 z, x y^2, xyz, x z^2, y^3, y^2 z, y z^2, z^3 | the quadratic list])
FP_HD int ll_slot(int i, int j) {          // linear x linear -> quadratic
    const int t[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
    return t[i][j];
}
FP_HD int ql_slot(int i, int j) {          // quadratic x linear -> cubic
    const int t[10][4] = {{0, 1, 2, 10}, {1, 3, 4, 11}, {2, 4, 5, 12}, {3, 6, 7, 13}, {4, 7, 8, 14}, {5, 8, 9, 15},
                          {10, 11, 12, 16}, {11, 13, 14, 17}, {12, 14, 15, 18}, {16, 17, 18, 19}};
    return t[i][j];
}
// out (registers) += sign * a * b for two linear polynomials = columns ca, cb of the basis
FP_HD void mul_ll(const double (&B)[4][9], int ca, int cb, double (&out)[10], double sign) {
    double a[4], b[4];
    FP_UNROLL for (int k = 0; k < 4; ++k) { a[k] = B[k][ca]; b[k] = B[k][cb]; }
    FP_UNROLL for (int i = 0; i < 4; ++i) {
        FP_UNROLL for (int j = 0; j < 4; ++j) out[ll_slot(i, j)] += sign * a[i] * b[j];
    }
}
// out (registers) += sign * a * b for a quadratic polynomial a[10] and the linear polynomial in column cb of the basis
FP_HD void mul_ql(const double* a10, const double (&B)[4][9], int cb, double (&out)[20], double sign) {
    double a[10], b[4];
    FP_UNROLL for (int k = 0; k < 10; ++k) a[k] = a10[k];
    FP_UNROLL for (int k = 0; k < 4; ++k) b[k] = B[k][cb];
    FP_UNROLL for (int i = 0; i < 10; ++i) {
        FP_UNROLL for (int j = 0; j < 4; ++j) out[ql_slot(i, j)] += sign * a[i] * b[j];
    }
}

// 4 vectors spanning the null space of the 5 x 9 system: Gauss-Jordan with full pivoting (largest |entry| of the remaining rows, first in
// row-major order on ties); free column f gives the vector with 1 at f and -R[i][f] at pivot column i.  Lane j owns column j.
template <int L>
FP_HD bool null_basis_5x9(double (&A)[5][9], double (&basis)[4][9], int lane) {
    unsigned piv = 0;                                     // pivot column of row r in bits 4 r .. 4 r + 3
    FP_LOOP for (int r = 0; r < 5; ++r) {
        double best = -1.0;
        int key = r * 9;
        FP_LOOP for (int t = lane; t < (5 - r) * 9; t += L) {
            const int i = r + t / 9, j = t - (t / 9) * 9;
            const double v = fabs(A[i][j]);
            if (v > best) { best = v; key = i * 9 + j; }
        }
        group_argmax<L>(best, key);
        if (!(best > 1e-14)) return false;
        const int pr = key / 9, pc = key - pr * 9;
        const double inv = 1.0 / A[pr][pc];               // one division per pivot, then products
        double f[5];                                       // column pc of the rows as they stand AFTER rows r and pr have changed places
        FP_UNROLL for (int i = 0; i < 5; ++i) f[i] = A[i == pr ? r : i][pc];
        group_fence<L>();
        FP_LOOP for (int j = lane; j < 9; j += L) {
            const double a = A[r][j], b = A[pr][j];
            const double row = b * inv;
            A[pr][j] = a;
            A[r][j] = row;
            FP_UNROLL for (int i = 0; i < 5; ++i) {
                const double v = A[i][j];
                A[i][j] = i == r ? v : v - f[i] * row;
            }
        }
        group_fence<L>();
        piv |= (unsigned)pc << (4 * r);
    }
    int nb = 0;
    FP_LOOP for (int f = 0; f < 9; ++f) {
        bool is_piv = false;
        FP_UNROLL for (int r = 0; r < 5; ++r) is_piv |= (int)((piv >> (4 * r)) & 15u) == f;
        if (is_piv) continue;
        if (nb < 4) {
            FP_LOOP for (int j = lane; j < 9; j += L) {
                double v = j == f ? 1.0 : 0.0;
                FP_UNROLL for (int r = 0; r < 5; ++r) v = (int)((piv >> (4 * r)) & 15u) == j ? -A[r][f] : v;
                basis[nb][j] = v;
            }
        }
        ++nb;
    }
    group_fence<L>();
    return nb == 4;
}

// [A1 | A2] -> [I | A1^-1 A2] by Gauss-Jordan with partial (row) pivoting, first largest on ties.  Lane j owns columns j, j + L, ...
template <int L>
FP_HD bool gauss_jordan_10x20(double (&A)[10][20], int lane) {
    FP_LOOP for (int c = 0; c < 10; ++c) {
        double best = -1.0;
        int pr = c;
        FP_LOOP for (int i = c + lane; i < 10; i += L) {
            const double v = fabs(A[i][c]);
            if (v > best) { best = v; pr = i; }
        }
        group_argmax<L>(best, pr);
        if (!(best > 1e-300)) return false;
        const double inv = 1.0 / A[pr][c];
        double f[10];                                      // column c of the rows as they stand AFTER rows c and pr have changed places
        FP_UNROLL for (int i = 0; i < 10; ++i) f[i] = A[i == pr ? c : i][c];
        group_fence<L>();
        FP_LOOP for (int j = lane; j < 20; j += L) {
            const double a = A[c][j], b = A[pr][j];
            const double row = b * inv;
            A[pr][j] = a;
            A[c][j] = row;
            double col[10];
            FP_UNROLL for (int i = 0; i < 10; ++i) col[i] = A[i][j];
            FP_UNROLL for (int i = 0; i < 10; ++i) A[i][j] = i == c ? col[i] : col[i] - f[i] * row;
        }
        group_fence<L>();
    }
    return true;
}

// reduction to upper Hessenberg form by stabilised elementary similarity transformations, then the eigenvalues by the shifted QR
// algorithm with implicit double shifts (EISPACK elmhes / hqr, 1-based indexing kept as published, n = 10); returns false if an
// eigenvalue needs more than 60 iterations.  The arithmetic is the published one, element for element; the row / column loops are
// dealt to the lanes, everything else is computed by every lane of the group (identical values).
template <int L>
FP_HD void hessenberg_10(double (&a)[11][11], int lane) {
    constexpr int n = 10;
    FP_LOOP for (int m = 2; m < n; ++m) {
        double best = 0.0;
        int i = m;
        FP_LOOP for (int j = m + lane; j <= n; j += L) {
            const double v = fabs(a[j][m - 1]);
            if (v > best) { best = v; i = j; }
        }
        group_argmax<L>(best, i);
        const double x = best > 0.0 ? a[i][m - 1] : 0.0;
        group_fence<L>();
        if (i != m) {
            FP_LOOP for (int j = m - 1 + lane; j <= n; j += L) { const double t = a[i][j]; a[i][j] = a[m][j]; a[m][j] = t; }
            group_fence<L>();
            FP_LOOP for (int j = 1 + lane; j <= n; j += L) { const double t = a[j][i]; a[j][i] = a[j][m]; a[j][m] = t; }
            group_fence<L>();
        }
        if (x != 0.0) {
            // The published loop treats the rows ii = m + 1 .. n one after the other (row operation, then column operation: two dependent
            // LDS round trips per row).  The elementary matrices L_ii = I - y_ii e_ii e_m^T of one step act on different rows and share
            // the column, so A <- (L_n .. L_m+1) A (L_m+1^-1 .. L_n^-1) can be applied as ALL row operations, then ALL column operations
            // (the same similarity transformation; only the order of a few roundings differs): three round trips per step.
            FP_LOOP for (int ii = m + 1 + lane; ii <= n; ii += L) a[ii][m - 1] = a[ii][m - 1] / x;      // the multipliers y_ii (stored as published)
            group_fence<L>();
            const int w = n - m + 1;                                       // columns m .. n
            FP_LOOP for (int t = lane; t < (n - m) * w; t += L) {
                const int r = t / w, ii = m + 1 + r, j = m + (t - r * w);
                a[ii][j] -= a[ii][m - 1] * a[m][j];
            }
            group_fence<L>();
            FP_LOOP for (int j = 1 + lane; j <= n; j += L) {
                double acc = a[j][m];
                FP_UNROLL for (int ii = 3; ii <= n; ++ii) acc += (ii > m ? a[ii][m - 1] : 0.0) * a[j][ii];      // (fixed trip count: the loads overlap)
                a[j][m] = acc;
            }
            group_fence<L>();
        }
    }
    FP_LOOP for (int i = 3 + lane; i <= n; i += L)
        for (int j = 1; j <= i - 2; ++j) a[i][j] = 0.0;
    group_fence<L>();
}

template <int L>
FP_HD bool eig_real_nonsym(double (&a)[11][11], double* wr, double* wi, int lane, unsigned long long* prof = nullptr) {
    (void)prof;
    constexpr int n = 10;
    hessenberg_10<L>(a, lane);
    double anorm = 0.0;
    for (int i = 1; i <= n; ++i)
        for (int j = (i - 1 > 1 ? i - 1 : 1); j <= n; ++j) anorm += fabs(a[i][j]);
    FP_STAMP(6);
    int nn = n;
    double t = 0.0, p = 0.0, q = 0.0, r = 0.0, s, x, y, z, w;
    while (nn >= 1) {
        int its = 0, l;
        do {
            {   // the largest l in [2, nn] with a negligible subdiagonal element a[l][l-1], else 1: the candidates are tested by the lanes
                double hit = 0.0;
                int key = 1;
                FP_LOOP for (int ll = 2 + lane; ll <= nn; ll += L) {
                    double ss = fabs(a[ll - 1][ll - 1]) + fabs(a[ll][ll]);
                    if (ss == 0.0) ss = anorm;
                    if (fabs(a[ll][ll - 1]) + ss == ss) { hit = (double)ll; key = ll; }
                }
                group_argmax<L>(hit, key);
                l = key;
                if (l >= 2) a[l][l - 1] = 0.0;
                group_fence<L>();
            }
            x = a[nn][nn];
            if (l == nn) {
                wr[nn] = x + t; wi[nn] = 0.0; --nn;
            } else {
                y = a[nn - 1][nn - 1];
                w = a[nn][nn - 1] * a[nn - 1][nn];
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = sqrt(fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + (p >= 0.0 ? fabs(z) : -fabs(z));
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn] = z; wi[nn - 1] = -z;
                    }
                    nn -= 2;
                } else {
                    if (its == 60) return false;
                    if (its == 10 || its == 20 || its == 40) {
                        t += x;
                        group_fence<L>();
                        FP_LOOP for (int i = 1 + lane; i <= nn; i += L) a[i][i] -= x;
                        group_fence<L>();
                        s = fabs(a[nn][nn - 1]) + fabs(a[nn - 1][nn - 2]);
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    FP_COUNT(7);
                    int m;
                    {   // start row of the double-shift sweep: the largest m in (l, nn - 2] whose subdiagonal coupling is negligible, else l;
                        // every candidate m with its own (p, q, r) on a lane, the winner's values broadcast
                        double hit = -1.0, pm = 0.0, qm = 0.0, rm = 0.0;
                        int key = l;
                        FP_LOOP for (int mm = l + lane; mm <= nn - 2; mm += L) {
                            const double zz = a[mm][mm];
                            double rr = x - zz, ss = y - zz;
                            double pp = (rr * ss - w) / a[mm + 1][mm] + a[mm][mm + 1];
                            double qq = a[mm + 1][mm + 1] - zz - rr - ss;
                            rr = a[mm + 2][mm + 1];
                            ss = fabs(pp) + fabs(qq) + fabs(rr);
                            pp /= ss; qq /= ss; rr /= ss;
                            bool take = mm == l;
                            if (!take) {
                                const double uu = fabs(a[mm][mm - 1]) * (fabs(qq) + fabs(rr));
                                const double vv = fabs(pp) * (fabs(a[mm - 1][mm - 1]) + fabs(zz) + fabs(a[mm + 1][mm + 1]));
                                take = uu + vv == vv;
                            }
                            if (take) { hit = (double)mm; key = mm; pm = pp; qm = qq; rm = rr; }
                        }
                        group_argmax<L>(hit, key);
                        m = key;
                        const int owner = (m - l) % L;
                        p = group_bcast<L>(pm, owner); q = group_bcast<L>(qm, owner); r = group_bcast<L>(rm, owner);
                        z = a[m][m];
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        a[i][i - 2] = 0.0;
                        if (i != m + 2) a[i][i - 3] = 0.0;
                    }
                    double pn = 0.0, qn = 0.0, rn = 0.0;               // the bulge column the next step starts from, taken from the lanes that wrote it
                    bool have_next = false;
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            if (have_next) { p = pn; q = qn; r = rn; }
                            else {
                                p = a[k][k - 1];
                                q = a[k + 1][k - 1];
                                r = 0.0;
                                if (k != nn - 1) r = a[k + 2][k - 1];
                            }
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { const double ix = 1.0 / x; p *= ix; q *= ix; r *= ix; }
                        }
                        have_next = false;
                        FP_COUNT(8);
                        const double nrm = sqrt(p * p + q * q + r * r);
                        s = p >= 0.0 ? nrm : -nrm;
                        if (s != 0.0) {
                            if (k == m) {
                                if (l != m) a[k][k - 1] = -a[k][k - 1];    // every lane reads, then every lane writes the same value
                            } else {
                                a[k][k - 1] = -s * x;
                            }
                            p += s;
                            {   // two reciprocals instead of five divisions (a division is ~12 dependent instructions)
                                const double is = 1.0 / s, ip = 1.0 / p;
                                x = p * is; y = q * is; z = r * is;
                                q *= ip; r *= ip;
                            }
                            const bool three = k != nn - 1;
                            group_fence<L>();
                            FP_LOOP for (int j = k + lane; j <= nn; j += L) {                 // row modification
                                double pp = a[k][j] + q * a[k + 1][j];
                                if (three) { pp += r * a[k + 2][j]; a[k + 2][j] -= pp * z; }
                                a[k + 1][j] -= pp * y;
                                a[k][j] -= pp * x;
                            }
                            group_fence<L>();
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            double mine = 0.0;                                               // this lane's new a[i][k] (a lane has at most one row: mmin - l < L)
                            FP_LOOP for (int i = l + lane; i <= mmin; i += L) {               // column modification
                                double pp = x * a[i][k] + y * a[i][k + 1];
                                if (three) { pp += z * a[i][k + 2]; a[i][k + 2] -= pp * r; }
                                a[i][k + 1] -= pp * q;
                                const double nv = a[i][k] - pp;
                                a[i][k] = nv;
                                if (L > 1) mine = nv;
                            }
                            group_fence<L>();
                            if (L > 1 && k + 1 <= nn - 1) {                                  // next step's p, q, r = a[k+1 .. k+3][k]
                                pn = group_bcast<L>(mine, (k + 1 - l) % L);
                                qn = group_bcast<L>(mine, (k + 2 - l) % L);
                                rn = k + 1 != nn - 1 ? group_bcast<L>(mine, (k + 3 - l) % L) : 0.0;
                                have_next = true;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return true;
}

// ---- the real eigenvalues as the real roots of the characteristic polynomial (round 4) ------------------------------------
// The shifted-QR iteration above is a chain of ~100 dependent bulge steps (2 700 cycles each on gfx950) however many lanes help.  Only the
// REAL eigenvalues are wanted, so: Hessenberg form (as before), its characteristic polynomial by the leading-block recurrence
//   p_k(l) = (l - h_kk) p_{k-1}(l) - sum_{i<k} h_ik (h_{i+1,i} ... h_{k,k-1}) p_{i-1}(l)          (coefficient j on lane j),
// and the real roots of the monic degree-10 polynomial by the derivative cascade: the roots of p^(m+1) cut the line into intervals on
// which p^(m) is monotonic, so each holds at most one root of p^(m), decided by the signs at its ends and found by a bracketed Newton
// iteration - one interval per lane, ten levels from the linear p^(9) down to p.  No deflation, no complex arithmetic, nothing that
// can fail to converge; close root pairs are found as long as fp64 still separates the signs (Nister's original solver isolates the
// roots of the same polynomial with Sturm sequences).  The scaled derivatives p^(m) / m! keep the coefficients within a factor 252.

// f = q(x), df = q'(x) for q = c[0] + c[1] x + ... + c[D] x^D
template <int D>
FP_HD void poly_eval(const double (&c)[11], double x, double& f, double& df) {
    double p = c[D], d = 0.0;
    FP_UNROLL for (int j = D - 1; j >= 0; --j) { d = d * x + p; p = p * x + c[j]; }
    f = p; df = d;
}

// 1 / x for a Newton step (the iteration corrects itself: the hardware estimate is enough on the GPU)
FP_HD double step_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    return 1.0 / x;
#endif
}

// the root of q in [a, b], q monotonic there, q(a) < 0 iff neg_a and the opposite sign at b: Newton steps, a bisection whenever a step
// would leave the bracket or stops halving the interval (the classic safeguarded iteration).  Relative accuracy 1e-11: these roots are
// separators for the next level or, at the last level, starting points of the correction on the matrix (hyman_polish)
template <int D>
FP_HD double root_in_bracket(const double (&c)[11], double a, double b, bool neg_a) {
    double lo = neg_a ? a : b, hi = neg_a ? b : a;        // q(lo) < 0 <= q(hi)
    double x = 0.5 * (a + b), dxold = fabs(b - a), dx = dxold, f, df;
    poly_eval<D>(c, x, f, df);
    if (f < 0.0) lo = x; else hi = x;
    FP_LOOP for (int it = 0; it < 200; ++it) {
        if (f == 0.0) return x;
        const bool bisect = !(((x - hi) * df - f) * ((x - lo) * df - f) < 0.0) || !(fabs(2.0 * f) <= fabs(dxold * df));
        dxold = dx;
        double xn;
        if (bisect) { dx = 0.5 * (hi - lo); xn = lo + dx; }
        else { dx = f * step_rcp(df); xn = x - dx; }
        if (xn == x) return x;
        x = xn;
        if (fabs(dx) <= 1e-11 * fabs(x)) return x;
        poly_eval<D>(c, x, f, df);
        if (f < 0.0) lo = x; else hi = x;
    }
    return x;
}

// characteristic polynomial of the upper Hessenberg matrix a (1-based, n = 10) -> P[10][0 .. 10] (monic)
template <int L>
FP_HD void charpoly_hessenberg(const double (&a)[11][11], double (&P)[11][12], int lane) {
    FP_LOOP for (int t = lane; t < 11 * 12; t += L) (&P[0][0])[t] = t == 0 ? 1.0 : 0.0;
    group_fence<L>();
    FP_UNROLL for (int k = 1; k <= 10; ++k) {
        FP_LOOP for (int j = lane; j <= k; j += L) {
            double acc = (j > 0 ? P[k - 1][j - 1] : 0.0) - a[k][k] * P[k - 1][j];
            double prod = 1.0;
            FP_UNROLL for (int i = k - 1; i >= 1; --i) {
                prod *= a[i + 1][i];
                acc -= a[i][k] * prod * P[i - 1][j];
            }
            P[k][j] = acc;
        }
        group_fence<L>();
    }
}

// Newton correction of an eigenvalue estimate on the MATRIX (Hyman's method: det(H - l I) of an upper Hessenberg matrix by back
// substitution of (H - l I) x = alpha e_1 with x_n = 1; the value is alpha times the subdiagonal product, which cancels in p / p').
// The coefficients of the characteristic polynomial carry the rounding of ~200 operations; its roots can be off by 1e-4 of an
// ill-conditioned solution, Hyman's evaluation is backward stable.  Steps larger than 1e-3 (1 + |l|) are refused (a root pair about to
// merge: the estimate stays).  a is 1-based, n = 10.
FP_HD double hyman_polish(const double (&a)[11][11], double lam) {
    constexpr int n = 10;
    FP_LOOP for (int it = 0; it < 2; ++it) {
        double x[n + 1], dx[n + 1];
        x[n] = 1.0; dx[n] = 0.0;
        bool ok = true;
        FP_UNROLL for (int i = n; i >= 2; --i) {
            double sx = 0.0, sd = 0.0;
            FP_UNROLL for (int j = n; j >= i; --j) {
                const double hij = j == i ? a[i][j] - lam : a[i][j];
                sx += hij * x[j]; sd += hij * dx[j];
            }
            sd -= x[i];
            const double sub = a[i][i - 1];
            ok = ok && sub != 0.0;
            const double inv = 1.0 / sub;
            x[i - 1] = -sx * inv; dx[i - 1] = -sd * inv;
        }
        double pv = 0.0, pd = 0.0;
        FP_UNROLL for (int j = n; j >= 1; --j) {
            const double hij = j == 1 ? a[1][j] - lam : a[1][j];
            pv += hij * x[j]; pd += hij * dx[j];
        }
        pd -= x[1];
        const double step = pv / pd;
        if (!ok || !isfinite(step) || !(fabs(step) <= 1e-3 * (1.0 + fabs(lam)))) break;
        lam -= step;
    }
    return lam;
}

// one level of the cascade: the real roots of q_m = p^(m) / m! (degree D = 10 - m) between the np separators in s.rt[(m + 1) & 1] (the
// real roots of q_(m+1)), ascending into s.rt[m & 1]; returns their number.  Interval i = (sep[i - 1], sep[i]) on lane i, the outer two open.
template <int L, int D>
FP_HD int root_level(RootScratch& s, int np, int lane) {
    constexpr int m = 10 - D;
    double c[11];
    FP_UNROLL for (int j = 0; j <= D; ++j) c[j] = s.Q[m][j];
    const double* sep = s.rt[(m + 1) & 1];
    FP_LOOP for (int i = lane; i <= np; i += L) {
        const bool open_a = i == 0, open_b = i == np;
        double a = open_a ? 0.0 : sep[i - 1], b = open_b ? 0.0 : sep[i];
        double fa = 0.0, fb = 0.0, dd;
        if (!open_a) poly_eval<D>(c, a, fa, dd);
        if (!open_b) poly_eval<D>(c, b, fb, dd);
        const bool neg_a = open_a ? (D & 1) != 0 : fa < 0.0;          // the leading coefficient C(10, m) is positive
        const bool neg_b = open_b ? false : fb < 0.0;
        bool has = neg_a != neg_b;
        double root = 0.0;
        if (has) {
            // an open end: walk outwards in doubling steps until the sign of the far side shows (the Cauchy bound guarantees it does)
            if (open_a) {
                const double from = open_b ? 0.0 : b;
                double step = fabs(from) > 1.0 ? fabs(from) : 1.0;
                int tries = 0;
                FP_LOOP for (;; ++tries) {
                    a = from - step;
                    poly_eval<D>(c, a, fa, dd);
                    if ((fa < 0.0) == neg_a || tries >= 400) break;
                    step *= 2.0;
                }
                has = has && tries < 400;
            }
            if (open_b) {
                const double from = open_a ? 0.0 : a;
                double step = fabs(from) > 1.0 ? fabs(from) : 1.0;
                int tries = 0;
                FP_LOOP for (;; ++tries) {
                    b = from + step;
                    poly_eval<D>(c, b, fb, dd);
                    if (!(fb < 0.0) || tries >= 400) break;
                    step *= 2.0;
                }
                has = has && tries < 400;
            }
            if (has) root = root_in_bracket<D>(c, a, b, neg_a);
            has = has && isfinite(root);
        }
        s.cand[i] = root;
        s.found[i] = has ? 1 : 0;
    }
    group_fence<L>();
    double* dst = s.rt[m & 1];
    int cnt = 0;
    FP_LOOP for (int i = 0; i <= np; ++i) {                // every lane: the same values to the same places
        if (s.found[i]) { dst[cnt] = s.cand[i]; ++cnt; }
    }
    group_fence<L>();
    return cnt;
}

// real roots of the monic polynomial c[0 .. 10] (c = s.P[10]), ascending, into out[]; returns their number (every lane: the same)
template <int L>
FP_HD int real_roots_monic10(RootScratch& s, double* out, int lane) {
    FP_LOOP for (int t = lane; t < 10 * 12; t += L) {
        const int m = t / 12, j = t - 12 * m;
        double v = 0.0;
        if (j + m <= 10) {
            double b = 1.0;                                // C(j + m, m), exact
            FP_LOOP for (int u = 1; u <= m; ++u) b = b * (double)(j + u) / (double)u;
            v = b * s.P[10][j + m];
        }
        s.Q[m][j] = v;
    }
    group_fence<L>();
    s.rt[1][0] = -s.Q[9][0] / s.Q[9][1];                   // level 9 is linear (every lane writes the same value)
    group_fence<L>();
    int np = 1;                                            // number of separators = real roots of the level above
    np = root_level<L, 2>(s, np, lane);
    np = root_level<L, 3>(s, np, lane);
    np = root_level<L, 4>(s, np, lane);
    np = root_level<L, 5>(s, np, lane);
    np = root_level<L, 6>(s, np, lane);
    np = root_level<L, 7>(s, np, lane);
    np = root_level<L, 8>(s, np, lane);
    np = root_level<L, 9>(s, np, lane);
    np = root_level<L, 10>(s, np, lane);
    FP_LOOP for (int i = lane; i < np; i += L) out[i] = s.rt[0][i];
    group_fence<L>();
    return np;
}

// Eigenvector of the action matrix M for a real eigenvalue lambda, i.e. the basis monomials b = [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
// at a solution: rows 6..9 of M are unit rows (x.x = x^2, x.y = xy, x.z = xz, x.1 = x), so with b9 = 1: b6 = lambda, b0 = lambda^2,
// b1 = lambda b7, b2 = lambda b8, and rows 0..5 of (M - lambda I) b = 0 are six linear equations in the five unknowns
// u = (b3, b4, b5, b7, b8) = (y^2, yz, z^2, y, z): a consistent 6 x 5 system, solved by elimination with full pivoting (no normal
// equations, which would square the condition number).  M[r][j] = -A[r][10 + j], the six non-trivial rows of the action matrix.
// One lane per eigenvalue (G = that lane's 6 x 6 scratch).  Returns (y, z).
FP_HD bool solve_yz(const double (&A)[10][20], double (&G)[6][6], double lam, double* y, double* z) {
    FP_LOOP for (int r = 0; r < 6; ++r) {
        double M[10];
        FP_UNROLL for (int j = 0; j < 10; ++j) M[j] = -A[r][10 + j];
        // (M - lam I)[r] . b = 0 with b = [lam^2, lam b7, lam b8, b3, b4, b5, lam, b7, b8, 1]
        const double m0 = M[0] - (r == 0 ? lam : 0.0), m1 = M[1] - (r == 1 ? lam : 0.0), m2 = M[2] - (r == 2 ? lam : 0.0);
        G[r][0] = M[3] - (r == 3 ? lam : 0.0);
        G[r][1] = M[4] - (r == 4 ? lam : 0.0);
        G[r][2] = M[5] - (r == 5 ? lam : 0.0);
        G[r][3] = M[7] + lam * m1;
        G[r][4] = M[8] + lam * m2;
        G[r][5] = -(lam * lam * m0 + lam * M[6] + M[9]);
    }
    int colperm[5] = {0, 1, 2, 3, 4};
    FP_LOOP for (int c = 0; c < 5; ++c) {
        int pr = c, pc = c;
        double best = -1.0;
        FP_LOOP for (int i = c; i < 6; ++i) {
            double v[5];
            FP_UNROLL for (int j = 0; j < 5; ++j) v[j] = fabs(G[i][j]);
            FP_UNROLL for (int j = 0; j < 5; ++j) {
                const bool gt = j >= c && v[j] > best;
                best = gt ? v[j] : best; pr = gt ? i : pr; pc = gt ? j : pc;
            }
        }
        if (!(best > 0.0)) return false;
        {   // row pr <-> row c, then column pc <-> column c (all rows)
            double rp[6], rc[6];
            FP_UNROLL for (int j = 0; j < 6; ++j) { rp[j] = G[pr][j]; rc[j] = G[c][j]; }
            FP_UNROLL for (int j = 0; j < 6; ++j) { G[pr][j] = rc[j]; }
            FP_UNROLL for (int j = 0; j < 6; ++j) { G[c][j] = rp[j]; }
            double cp[6], cc[6];
            FP_UNROLL for (int i = 0; i < 6; ++i) { cp[i] = G[i][pc]; cc[i] = G[i][c]; }
            FP_UNROLL for (int i = 0; i < 6; ++i) { G[i][pc] = cc[i]; }
            FP_UNROLL for (int i = 0; i < 6; ++i) { G[i][c] = cp[i]; }
            int v = colperm[0], u = colperm[0];
            FP_UNROLL for (int j = 1; j < 5; ++j) { v = pc == j ? colperm[j] : v; u = c == j ? colperm[j] : u; }
            FP_UNROLL for (int j = 0; j < 5; ++j) colperm[j] = j == c ? v : (j == pc ? u : colperm[j]);
        }
        double row[6];
        FP_UNROLL for (int j = 0; j < 6; ++j) row[j] = G[c][j];
        const double inv = 1.0 / G[c][c];
        FP_UNROLL for (int j = 0; j < 6; ++j) row[j] = j >= c ? row[j] * inv : row[j];
        FP_UNROLL for (int j = 0; j < 6; ++j) G[c][j] = row[j];
        FP_LOOP for (int i = 0; i < 6; ++i) {
            if (i != c) {
                const double f = G[i][c];
                double ri[6];
                FP_UNROLL for (int j = 0; j < 6; ++j) ri[j] = G[i][j];
                FP_UNROLL for (int j = 0; j < 6; ++j) G[i][j] = j >= c ? ri[j] - f * row[j] : ri[j];
            }
        }
    }
    double yy = 0.0, zz = 0.0;
    FP_UNROLL for (int c = 0; c < 5; ++c) { yy = colperm[c] == 3 ? G[c][5] : yy; zz = colperm[c] == 4 ? G[c][5] : zz; }
    *y = yy;
    *z = zz;
    return isfinite(yy) && isfinite(zz);
}


// w.pts: 5 normalised correspondences (x1h^T E x0h = 0), written by the caller (all lanes of the group see them).  Writes up to 10
// essential matrices (row-major, Frobenius norm 1; Eout [10][9]) in ascending order of the eigenvalue and returns their number (the same
// value on every lane of the group).
template <int L>
FP_HD int five_point(double* Eout, Work& w, int lane, unsigned long long* prof = nullptr) {
    (void)prof;
    FP_STAMP(0);
    FP_LOOP for (int i = lane; i < 5; i += L) {
        const double ax = w.pts[i][0], ay = w.pts[i][1], bx = w.pts[i][2], by = w.pts[i][3];
        const double row[9] = {bx * ax, bx * ay, bx, by * ax, by * ay, by, ax, ay, 1.0};
        FP_UNROLL for (int j = 0; j < 9; ++j) w.Q[i][j] = row[j];
    }
    group_fence<L>();
    double (&B)[4][9] = w.B;
    if (!null_basis_5x9<L>(w.Q, B, lane)) return 0;
    FP_STAMP(1);
    // entry (i, j) of E = column 3 i + j of the basis.  Jobs 0..2: the 2 x 2 minors of rows 1, 2 (det(E) by the first row; minor t leaves
    // out column t); jobs 3..11: the entries of E E^T
    FP_LOOP for (int job = lane; job < 12; job += L) {
        double m[10];
        FP_UNROLL for (int k = 0; k < 10; ++k) m[k] = 0.0;
        if (job < 3) {
            const int c1 = job == 0 ? 1 : 0, c2 = job == 2 ? 1 : 2;
            mul_ll(B, 3 + c1, 6 + c2, m, 1.0);
            mul_ll(B, 3 + c2, 6 + c1, m, -1.0);
            FP_UNROLL for (int k = 0; k < 10; ++k) w.minors[job][k] = m[k];
        } else {
            const int ij = job - 3, i = ij / 3, j = ij - 3 * i;
            FP_LOOP for (int k = 0; k < 3; ++k) mul_ll(B, 3 * i + k, 3 * j + k, m, 1.0);
            FP_UNROLL for (int k = 0; k < 10; ++k) w.EEt[ij][k] = m[k];
        }
    }
    group_fence<L>();
    FP_LOOP for (int k = lane; k < 10; k += L) w.tr[k] = w.EEt[0][k] + w.EEt[4][k] + w.EEt[8][k];
    group_fence<L>();
    // the ten cubics: det(E) = 0 and the nine entries of 2 E E^T E - tr(E E^T) E = 0; a row per lane
    FP_LOOP for (int r = lane; r < 10; r += L) {
        double row[20];
        FP_UNROLL for (int k = 0; k < 20; ++k) row[k] = 0.0;
        const int i = r == 0 ? 0 : (r - 1) / 3, j = r == 0 ? 0 : (r - 1) - 3 * i;
        FP_LOOP for (int k = 0; k < (r == 0 ? 3 : 4); ++k) {
            const double* a = r == 0 ? w.minors[k] : (k < 3 ? w.EEt[3 * i + k] : w.tr);
            const int cb = r == 0 ? k : (k < 3 ? 3 * k + j : 3 * i + j);
            const double sign = r == 0 ? (k == 1 ? -1.0 : 1.0) : (k < 3 ? 2.0 : -1.0);
            mul_ql(a, B, cb, row, sign);
        }
        FP_UNROLL for (int k = 0; k < 20; ++k) w.A[r][k] = row[k];
    }
    group_fence<L>();
    FP_STAMP(2);
    if (!gauss_jordan_10x20<L>(w.A, lane)) return 0;
    FP_STAMP(3);
    // multiplication by x on [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]: x . (first six) = the degree-3 monomials x^3, x^2 y, x^2 z, x y^2, xyz, x z^2
    FP_LOOP for (int t = lane; t < 100; t += L) {
        const int i = t / 10, j = t - 10 * i;
        const bool unit = (i == 6 && j == 0) || (i == 7 && j == 1) || (i == 8 && j == 2) || (i == 9 && j == 6);
        w.H[i + 1][j + 1] = i < 6 ? -w.A[i][10 + j] : (unit ? 1.0 : 0.0);
    }
    group_fence<L>();
    int nl = 0;
#ifdef FP_EIG_QR                                           // rounds 2-3: every eigenvalue by the shifted QR iteration (kept for A/B runs)
    if (!eig_real_nonsym<L>(w.H, w.wr, w.wi, lane, prof)) return 0;
    FP_STAMP(4);
    // the real eigenvalues in ascending order, in place: wr[0 .. nl) (step i writes below index i and reads index i and above); every lane
    FP_LOOP for (int i = 1; i <= 10; ++i) {
        const double re = w.wr[i], im = w.wi[i];
        if (im == 0.0 && isfinite(re)) {
            int k = nl++;
            while (k > 0 && w.wr[k - 1] > re) { w.wr[k] = w.wr[k - 1]; --k; }
            w.wr[k] = re;
        }
    }
    group_fence<L>();
#else
    hessenberg_10<L>(w.H, lane);
    FP_STAMP(6);
    charpoly_hessenberg<L>(w.H, w.rs.P, lane);
    FP_STAMP(7);
    nl = real_roots_monic10<L>(w.rs, w.wr, lane);
    FP_LOOP for (int e = lane; e < nl; e += L) w.wr[e] = hyman_polish(w.H, w.wr[e]);
    group_fence<L>();
    FP_LOOP for (int i = 1; i < nl; ++i) {                 // ascending order again, should two close roots have crossed (every lane, same values)
        const double v = w.wr[i];
        int k = i;
        while (k > 0 && w.wr[k - 1] > v) { w.wr[k] = w.wr[k - 1]; --k; }
        w.wr[k] = v;
    }
    group_fence<L>();
    FP_STAMP(4);
#endif
    FP_LOOP for (int e = lane; e < nl; e += L) {          // an eigenvalue per lane
        const double x = w.wr[e];
        double y, z;
        bool ok = solve_yz(w.A, w.G[e], x, &y, &z);
        double nrm = 0.0, Es[9], b0[9], b1[9], b2[9], b3[9];
        FP_UNROLL for (int k = 0; k < 9; ++k) { b0[k] = B[0][k]; b1[k] = B[1][k]; b2[k] = B[2][k]; b3[k] = B[3][k]; }
        FP_UNROLL for (int k = 0; k < 9; ++k) { Es[k] = ok ? x * b0[k] + y * b1[k] + z * b2[k] + b3[k] : 0.0; nrm += Es[k] * Es[k]; }
        nrm = sqrt(nrm);
        ok = ok && nrm > 0.0 && isfinite(nrm);
        FP_UNROLL for (int k = 0; k < 9; ++k) w.Es[e][k] = Es[k] / nrm;
        w.ok[e] = ok ? 1 : 0;
    }
    group_fence<L>();
    int nout = 0;
    FP_LOOP for (int e = 0; e < nl; ++e) {
        if (!w.ok[e]) continue;
        FP_LOOP for (int k = lane; k < 9; k += L) Eout[nout * 9 + k] = w.Es[e][k];
        ++nout;
    }
    FP_STAMP(5);
    return nout;
}

}  // namespace fivept
