// Five-point relative pose (minimal solver of cv2.findEssentialMat, eval/pose_estimation.py:96-105), restated from the publications -
// D. Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004, in the formulation of H. Stewenius, C. Engels,
// D. Nister, "Recent developments on direct relative orientation", ISPRS J. 2006:
//   E = x X + y Y + z Z + W in the 4-dimensional null space of the five epipolar constraints; det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0
//   are ten cubics in (x, y, z); eliminating the ten degree-3 monomials leaves the multiplication-by-x map on the quotient-ring basis
//   [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1] as a 10 x 10 matrix whose real eigenpairs are the (up to ten) solutions.
// Plain fp64 scalar code, one hypothesis per GPU thread; also compiled for the host by the unit test of the algebra (tests/test_pose.py
// builds tools/probe/fivept_host.cpp with g++).  CPU twin: oracle/pose_oracle.py five_point (same elimination order, same candidate order).
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define FP_HD __host__ __device__
#else
#define FP_HD
#endif

namespace fivept {

// Scratch of one solve.  On the GPU a thread's private arrays with run-time indices live in scratch memory (an L2 round trip per access;
// the solver is one long dependent chain, ~25 k such accesses: 1.8 ms per call); the kernel therefore hands every thread a Work in LDS.
struct Work {
    double Q[5][9], B[4][9], E[3][3][4], A[10][20], EEt[3][3][10], tr[10], m[10], M[10][10], H[11][11], wr[11], wi[11], lam[10];
    double C[6][5], d[6], G[6][6];
};

// slot tables of the polynomial products (monomial orders: linear [x, y, z, 1]; quadratic [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1];
// cubic [x^3, x^2 y, x^2 z, x y^2, xyz, x z^2, y^3, y^2 z, y z^2, z^3 | the quadratic list])
FP_HD inline int ll_slot(int i, int j) {          // linear x linear -> quadratic
    const int t[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
    return t[i][j];
}
FP_HD inline int ql_slot(int i, int j) {          // quadratic x linear -> cubic
    const int t[10][4] = {{0, 1, 2, 10}, {1, 3, 4, 11}, {2, 4, 5, 12}, {3, 6, 7, 13}, {4, 7, 8, 14}, {5, 8, 9, 15},
                          {10, 11, 12, 16}, {11, 13, 14, 17}, {12, 14, 15, 18}, {16, 17, 18, 19}};
    return t[i][j];
}
FP_HD inline void mul_ll(const double* a, const double* b, double* out, double sign, bool accumulate) {
    if (!accumulate) for (int k = 0; k < 10; ++k) out[k] = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[ll_slot(i, j)] += sign * a[i] * b[j];
}
FP_HD inline void mul_ql(const double* a, const double* b, double* out, double sign) {      // accumulates
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) out[ql_slot(i, j)] += sign * a[i] * b[j];
}

// 4 vectors spanning the null space of the 5 x 9 system: Gauss-Jordan with full pivoting (largest |entry| of the remaining rows, first in
// row-major order on ties); free column f gives the vector with 1 at f and -R[i][f] at pivot column i
FP_HD inline bool null_basis_5x9(double (&A)[5][9], double (&basis)[4][9]) {
    int piv[5];
    for (int r = 0; r < 5; ++r) {
        int pr = r, pc = 0;
        double best = -1.0;
        for (int i = r; i < 5; ++i)
            for (int j = 0; j < 9; ++j)
                if (fabs(A[i][j]) > best) { best = fabs(A[i][j]); pr = i; pc = j; }
        if (!(best > 1e-14)) return false;
        if (pr != r) for (int j = 0; j < 9; ++j) { const double t = A[r][j]; A[r][j] = A[pr][j]; A[pr][j] = t; }
        const double d = A[r][pc];
        for (int j = 0; j < 9; ++j) A[r][j] /= d;
        for (int i = 0; i < 5; ++i)
            if (i != r) {
                const double f = A[i][pc];
                for (int j = 0; j < 9; ++j) A[i][j] -= f * A[r][j];
            }
        piv[r] = pc;
    }
    int nb = 0;
    for (int f = 0; f < 9; ++f) {
        bool is_piv = false;
        for (int r = 0; r < 5; ++r) is_piv |= piv[r] == f;
        if (is_piv) continue;
        for (int j = 0; j < 9; ++j) basis[nb][j] = 0.0;
        basis[nb][f] = 1.0;
        for (int r = 0; r < 5; ++r) basis[nb][piv[r]] = -A[r][f];
        ++nb;
    }
    return nb == 4;
}

// reduction to upper Hessenberg form by stabilised elementary similarity transformations, then the eigenvalues by the shifted QR
// algorithm with implicit double shifts (EISPACK elmhes / hqr, 1-based indexing kept as published); returns false if an eigenvalue
// needs more than 60 iterations
FP_HD inline bool eig_real_nonsym(double (&a)[11][11], int n, double* wr, double* wi) {
    for (int m = 2; m < n; ++m) {
        double x = 0.0;
        int i = m;
        for (int j = m; j <= n; ++j)
            if (fabs(a[j][m - 1]) > fabs(x)) { x = a[j][m - 1]; i = j; }
        if (i != m) {
            for (int j = m - 1; j <= n; ++j) { const double t = a[i][j]; a[i][j] = a[m][j]; a[m][j] = t; }
            for (int j = 1; j <= n; ++j) { const double t = a[j][i]; a[j][i] = a[j][m]; a[j][m] = t; }
        }
        if (x != 0.0)
            for (i = m + 1; i <= n; ++i) {
                double y = a[i][m - 1];
                if (y != 0.0) {
                    y /= x;
                    a[i][m - 1] = y;
                    for (int j = m; j <= n; ++j) a[i][j] -= y * a[m][j];
                    for (int j = 1; j <= n; ++j) a[j][m] += y * a[j][i];
                }
            }
    }
    for (int i = 3; i <= n; ++i)
        for (int j = 1; j <= i - 2; ++j) a[i][j] = 0.0;
    double anorm = 0.0;
    for (int i = 1; i <= n; ++i)
        for (int j = (i - 1 > 1 ? i - 1 : 1); j <= n; ++j) anorm += fabs(a[i][j]);
    int nn = n;
    double t = 0.0, p = 0.0, q = 0.0, r = 0.0, s, x, y, z, w, u, v;
    while (nn >= 1) {
        int its = 0, l;
        do {
            for (l = nn; l >= 2; --l) {
                s = fabs(a[l - 1][l - 1]) + fabs(a[l][l]);
                if (s == 0.0) s = anorm;
                if (fabs(a[l][l - 1]) + s == s) { a[l][l - 1] = 0.0; break; }
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
                        for (int i = 1; i <= nn; ++i) a[i][i] -= x;
                        s = fabs(a[nn][nn - 1]) + fabs(a[nn - 1][nn - 2]);
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = a[m][m];
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / a[m + 1][m] + a[m][m + 1];
                        q = a[m + 1][m + 1] - z - r - s;
                        r = a[m + 2][m + 1];
                        s = fabs(p) + fabs(q) + fabs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        u = fabs(a[m][m - 1]) * (fabs(q) + fabs(r));
                        v = fabs(p) * (fabs(a[m - 1][m - 1]) + fabs(z) + fabs(a[m + 1][m + 1]));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        a[i][i - 2] = 0.0;
                        if (i != m + 2) a[i][i - 3] = 0.0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = a[k][k - 1];
                            q = a[k + 1][k - 1];
                            r = 0.0;
                            if (k != nn - 1) r = a[k + 2][k - 1];
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        const double nrm = sqrt(p * p + q * q + r * r);
                        s = p >= 0.0 ? nrm : -nrm;
                        if (s != 0.0) {
                            if (k == m) {
                                if (l != m) a[k][k - 1] = -a[k][k - 1];
                            } else {
                                a[k][k - 1] = -s * x;
                            }
                            p += s;
                            x = p / s; y = q / s; z = r / s;
                            q /= p; r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = a[k][j] + q * a[k + 1][j];
                                if (k != nn - 1) { p += r * a[k + 2][j]; a[k + 2][j] -= p * z; }
                                a[k + 1][j] -= p * y;
                                a[k][j] -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * a[i][k] + y * a[i][k + 1];
                                if (k != nn - 1) { p += z * a[i][k + 2]; a[i][k + 2] -= p * r; }
                                a[i][k + 1] -= p * q;
                                a[i][k] -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return true;
}

// Eigenvector of the action matrix M for a real eigenvalue lambda, i.e. the basis monomials b = [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
// at a solution: rows 6..9 of M are unit rows (x.x = x^2, x.y = xy, x.z = xz, x.1 = x), so with b9 = 1: b6 = lambda, b0 = lambda^2,
// b1 = lambda b7, b2 = lambda b8, and rows 0..5 of (M - lambda I) b = 0 are six linear equations in the five unknowns
// u = (b3, b4, b5, b7, b8) = (y^2, yz, z^2, y, z): a consistent 6 x 5 system, solved by elimination with full pivoting.  ~200 operations instead of a 10 x 10 elimination per eigenvalue.  Returns (y, z).
FP_HD inline bool solve_yz(Work& w, double lam, double* y, double* z) {
    double (&M)[10][10] = w.M;
    double (&C)[6][5] = w.C;
    double (&d)[6] = w.d;
    for (int r = 0; r < 6; ++r) {
        // (M - lam I)[r] . b = 0 with b = [lam^2, lam b7, lam b8, b3, b4, b5, lam, b7, b8, 1]
        const double m0 = M[r][0] - (r == 0 ? lam : 0.0), m1 = M[r][1] - (r == 1 ? lam : 0.0), m2 = M[r][2] - (r == 2 ? lam : 0.0);
        C[r][0] = M[r][3] - (r == 3 ? lam : 0.0);
        C[r][1] = M[r][4] - (r == 4 ? lam : 0.0);
        C[r][2] = M[r][5] - (r == 5 ? lam : 0.0);
        C[r][3] = M[r][7] + lam * m1;
        C[r][4] = M[r][8] + lam * m2;
        d[r] = -(lam * lam * m0 + lam * M[r][6] + M[r][9]);
    }
    // the system is consistent (rank 5): Gaussian elimination with full pivoting over the 6 rows picks five of them (no normal equations,
    // which would square the condition number)
    double (&G)[6][6] = w.G;
    int colperm[5] = {0, 1, 2, 3, 4};
    for (int r = 0; r < 6; ++r) { for (int j = 0; j < 5; ++j) G[r][j] = C[r][j]; G[r][5] = d[r]; }
    for (int c = 0; c < 5; ++c) {
        int pr = c, pc = c;
        double best = -1.0;
        for (int i = c; i < 6; ++i)
            for (int j = c; j < 5; ++j)
                if (fabs(G[i][j]) > best) { best = fabs(G[i][j]); pr = i; pc = j; }
        if (!(best > 0.0)) return false;
        if (pr != c) for (int j = 0; j < 6; ++j) { const double t = G[c][j]; G[c][j] = G[pr][j]; G[pr][j] = t; }
        if (pc != c) {
            for (int i = 0; i < 6; ++i) { const double t = G[i][c]; G[i][c] = G[i][pc]; G[i][pc] = t; }
            const int t = colperm[c]; colperm[c] = colperm[pc]; colperm[pc] = t;
        }
        const double piv = G[c][c];
        for (int j = c; j < 6; ++j) G[c][j] /= piv;
        for (int i = 0; i < 6; ++i)
            if (i != c) {
                const double f = G[i][c];
                if (f != 0.0) for (int j = c; j < 6; ++j) G[i][j] -= f * G[c][j];
            }
    }
    double u[5];
    for (int c = 0; c < 5; ++c) u[colperm[c]] = G[c][5];
    *y = u[3];
    *z = u[4];
    return isfinite(*y) && isfinite(*z);
}

// x0, x1: 5 normalised correspondences (x1h^T E x0h = 0).  Writes up to 10 essential matrices (row-major, Frobenius norm 1; Eout [10][9])
// in ascending order of the eigenvalue and returns their number
FP_HD inline int five_point(const double (&x0)[5][2], const double (&x1)[5][2], double* Eout, Work& w) {
    double (&Q)[5][9] = w.Q;
    for (int i = 0; i < 5; ++i) {
        const double ax = x0[i][0], ay = x0[i][1], bx = x1[i][0], by = x1[i][1];
        const double row[9] = {bx * ax, bx * ay, bx, by * ax, by * ay, by, ax, ay, 1.0};
        for (int j = 0; j < 9; ++j) Q[i][j] = row[j];
    }
    double (&B)[4][9] = w.B;
    if (!null_basis_5x9(Q, B)) return 0;
    double (&E)[3][3][4] = w.E;                                            // entries as linear polynomials [x, y, z, 1]
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 4; ++k) E[i][j][k] = B[k][3 * i + j];
    double (&A)[10][20] = w.A;
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 20; ++j) A[i][j] = 0.0;
    {   // det(E) by the first row
        double (&m)[10] = w.m;
        mul_ll(E[1][1], E[2][2], m, 1.0, false); mul_ll(E[1][2], E[2][1], m, -1.0, true); mul_ql(m, E[0][0], A[0], 1.0);
        mul_ll(E[1][0], E[2][2], m, 1.0, false); mul_ll(E[1][2], E[2][0], m, -1.0, true); mul_ql(m, E[0][1], A[0], -1.0);
        mul_ll(E[1][0], E[2][1], m, 1.0, false); mul_ll(E[1][1], E[2][0], m, -1.0, true); mul_ql(m, E[0][2], A[0], 1.0);
    }
    double (&EEt)[3][3][10] = w.EEt;
    double (&tr)[10] = w.tr;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            mul_ll(E[i][0], E[j][0], EEt[i][j], 1.0, false);
            mul_ll(E[i][1], E[j][1], EEt[i][j], 1.0, true);
            mul_ll(E[i][2], E[j][2], EEt[i][j], 1.0, true);
        }
    for (int k = 0; k < 10; ++k) tr[k] = EEt[0][0][k] + EEt[1][1][k] + EEt[2][2][k];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double* row = A[1 + 3 * i + j];
            for (int k = 0; k < 3; ++k) mul_ql(EEt[i][k], E[k][j], row, 2.0);
            mul_ql(tr, E[i][j], row, -1.0);
        }
    // [A1 | A2] -> A1^-1 A2 by Gauss-Jordan with partial pivoting
    for (int c = 0; c < 10; ++c) {
        int pr = c;
        for (int i = c + 1; i < 10; ++i) if (fabs(A[i][c]) > fabs(A[pr][c])) pr = i;
        if (!(fabs(A[pr][c]) > 1e-300)) return 0;
        if (pr != c) for (int j = 0; j < 20; ++j) { const double t = A[c][j]; A[c][j] = A[pr][j]; A[pr][j] = t; }
        const double d = A[c][c];
        for (int j = c; j < 20; ++j) A[c][j] /= d;
        for (int i = 0; i < 10; ++i)
            if (i != c) {
                const double f = A[i][c];
                if (f != 0.0) for (int j = c; j < 20; ++j) A[i][j] -= f * A[c][j];
            }
    }
    // multiplication by x on [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]: x . (first six) = the degree-3 monomials x^3, x^2 y, x^2 z, x y^2, xyz, x z^2
    double (&M)[10][10] = w.M;
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) M[i][j] = i < 6 ? -A[i][10 + j] : 0.0;
    M[6][0] = M[7][1] = M[8][2] = M[9][6] = 1.0;
    double (&H)[11][11] = w.H;
    double (&wr)[11] = w.wr;
    double (&wi)[11] = w.wi;
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) H[i + 1][j + 1] = M[i][j];
    if (!eig_real_nonsym(H, 10, wr, wi)) return 0;
    double (&lam)[10] = w.lam;
    int nl = 0;
    for (int i = 1; i <= 10; ++i)
        if (wi[i] == 0.0 && isfinite(wr[i])) {                     // insertion sort, ascending
            int k = nl++;
            while (k > 0 && lam[k - 1] > wr[i]) { lam[k] = lam[k - 1]; --k; }
            lam[k] = wr[i];
        }
    int nout = 0;
    for (int e = 0; e < nl; ++e) {
        double y, z;
        if (!solve_yz(w, lam[e], &y, &z)) continue;
        const double x = lam[e];
        double nrm = 0.0, Es[9];
        for (int k = 0; k < 9; ++k) { Es[k] = x * B[0][k] + y * B[1][k] + z * B[2][k] + B[3][k]; nrm += Es[k] * Es[k]; }
        nrm = sqrt(nrm);
        if (!(nrm > 0.0) || !isfinite(nrm)) continue;
        for (int k = 0; k < 9; ++k) Eout[nout * 9 + k] = Es[k] / nrm;
        ++nout;
    }
    return nout;
}

}  // namespace fivept
