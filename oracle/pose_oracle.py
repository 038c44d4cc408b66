"""CPU ORACLE for the pose step of the iterative loops.   *** TEST INFRASTRUCTURE ONLY ***

Two parts with different pin status:

* ``decompose_essential_mat`` - the 4-way cheirality vote of eval/pose_estimation.py:13-89, restated in numpy (fp64).  The two
  OpenCV primitives it calls are restated from their published algorithms: ``cv2.decomposeEssentialMat`` (SVD, W-matrix
  construction, OpenCV calib3d/five-point.cpp) and ``cv2.triangulatePoints`` (per-point DLT: null vector of the 4x4 system).
  ``cv2`` is absent from this image, so the restatement is pinned by construction properties only (tests/test_pose.py: the
  true pose wins the vote, masks follow the geometric definition); **parity with cv2 itself is unpinned**.
* ``estimate_pose`` - a seeded 8-point RANSAC with Sampson-distance inliers.  The reference calls
  ``cv2.findEssentialMat(method=USAC_MAGSAC)`` (eval/pose_estimation.py:96-105, opencv-contrib-python 4.5.5.64): a
  third-party randomized solver with no golden vectors in the reference - **MAGSAC parity is unpinned and not claimed**.
  This function is the CPU twin of csrc/pose.hip (same hypothesis sampling, same algebra) used to check the GPU kernels.
"""
from __future__ import annotations

import numpy as np

W_MAT = np.array([[0., 1., 0.], [-1., 0., 0.], [0., 0., 1.]])


def decompose_E(E):
    """cv2.decomposeEssentialMat: E = U diag(s) V^T, det(U), det(V^T) forced positive, R1 = U W V^T, R2 = U W^T V^T, t = U[:, 2]"""
    U, _, Vt = np.linalg.svd(np.asarray(E, dtype=np.float64))
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    return U @ W_MAT @ Vt, U @ W_MAT.T @ Vt, U[:, 2].copy()


def triangulate(P0, P1, x0, x1):
    """cv2.triangulatePoints (DLT): x0, x1 [2, n] normalised image points -> homogeneous [4, n]"""
    n = x0.shape[1]
    out = np.zeros((4, n))
    for i in range(n):
        A = np.stack([x0[0, i] * P0[2] - P0[0], x0[1, i] * P0[2] - P0[1],
                      x1[0, i] * P1[2] - P1[0], x1[1, i] * P1[2] - P1[1]])
        out[:, i] = np.linalg.svd(A)[2][-1]
    return out


def _mask_from_pts4d(Q, P, distance_thresh):
    """eval/pose_estimation.py:14-27"""
    Q = Q.copy()
    mask = (Q[2] * Q[3]) > 0
    Q = Q / Q[3]
    mask = (Q[2] < distance_thresh) & mask
    Qc = P @ Q
    mask = (Qc[2] > 0) & mask
    mask = (Qc[2] < distance_thresh) & mask
    return mask


def decompose_essential_mat(E, pts0, pts1, K0, K1, distance_thresh=1000):
    """eval/pose_estimation.py:13-89 -> (R, t, mask): the candidate (R1|R2, +-t) with most points in front of both cameras;
    ties resolved in the reference's order (R1,t), (R2,t), (R1,-t), (R2,-t).  pts in pixels; K = (K0 + K1) / 2 (:29)."""
    K = (np.asarray(K0, dtype=np.float64) + np.asarray(K1, dtype=np.float64)) / 2.
    p0 = np.asarray(pts0, dtype=np.float64).copy()
    p1 = np.asarray(pts1, dtype=np.float64).copy()
    for p in (p0, p1):
        p[:, 0] = (p[:, 0] - K[0, 2]) / K[0, 0]
        p[:, 1] = (p[:, 1] - K[1, 2]) / K[1, 1]
    x0, x1 = p0.T, p1.T
    R1, R2, t = decompose_E(E)
    P0 = np.eye(3, 4)
    cands = [(R1, t), (R2, t), (R1, -t), (R2, -t)]
    masks = []
    for R, tt in cands:
        P = np.concatenate([R, tt[:, None]], axis=1)
        masks.append(_mask_from_pts4d(triangulate(P0, P, x0, x1), P, distance_thresh))
    good = [int(m.sum()) for m in masks]
    best = max(good)
    for (R, tt), m, g in zip(cands, masks, good):
        if g == best:
            return R, tt, m
    raise AssertionError


# ------------------------------------------------------------------------------------------------ seeded 8-point RANSAC
def sample_index(seed: int, h: int, k: int, n: int) -> int:
    """k-th correspondence of hypothesis h: a 32-bit integer hash of (seed, h, k) modulo n (csrc/pose.hip pose_rand)"""
    x = (seed * 0x9E3779B1 + h * 0x85EBCA77 + k * 0xC2B2AE3D + 0x27D4EB2F) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12; x = (x * 0x297A2D39) & 0xFFFFFFFF
    x ^= x >> 15
    return x % n


def normalise(kpts, K):
    K = np.asarray(K, dtype=np.float64)
    k = np.asarray(kpts, dtype=np.float64)
    return np.stack([(k[:, 0] - K[0, 2]) / K[0, 0], (k[:, 1] - K[1, 2]) / K[1, 1]], axis=1)


def essential_from(x0, x1):
    """8-point algorithm on normalised points with Hartley conditioning, projected onto the essential manifold (1, 1, 0)"""
    def cond(x):
        c = x.mean(0)
        s = np.sqrt(2.0) / max(np.sqrt(((x - c) ** 2).sum(1)).mean(), 1e-12)
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.]])
        return (x - c) * s, T
    a, T0 = cond(x0)
    b, T1 = cond(x1)
    A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                  a[:, 0], a[:, 1], np.ones(len(a))], axis=1)
    w, V = np.linalg.eigh(A.T @ A)
    F = V[:, 0].reshape(3, 3)
    E = T1.T @ F @ T0
    U, _, Vt = np.linalg.svd(E)
    return U @ np.diag([1., 1., 0.]) @ Vt


def sampson_sq(E, x0, x1):
    h0 = np.concatenate([x0, np.ones((len(x0), 1))], 1)
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    Ex0 = h0 @ E.T
    Etx1 = h1 @ E
    num = (h1 * Ex0).sum(1) ** 2
    den = Ex0[:, 0] ** 2 + Ex0[:, 1] ** 2 + Etx1[:, 0] ** 2 + Etx1[:, 1] ** 2
    return num / np.maximum(den, 1e-30)


# ------------------------------------------------------------------------------------------------ five-point minimal solver
# Nister's problem in Stewenius' formulation (H. Stewenius, C. Engels, D. Nister, "Recent developments on direct relative orientation",
# ISPRS J. 2006): E lives in the 4-dimensional null space of the 5 epipolar constraints, E = x X + y Y + z Z + W; det(E) = 0 and
# 2 E E^T E - tr(E E^T) E = 0 are ten cubics in (x, y, z); after elimination of the ten degree-3 monomials the multiplication-by-x map
# on the quotient-ring basis [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1] is a 10 x 10 matrix whose real eigenpairs are the solutions.
# This is the minimal solver inside cv2.findEssentialMat; restated here from the publication (cv2 is absent), twin of csrc/pose.hip.
_MON3 = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
         (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_MON2 = _MON3[10:]                                   # quadratic polynomials: 10 coefficients in this order
_MON1 = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_I3 = {m: i for i, m in enumerate(_MON3)}
_I2 = {m: i for i, m in enumerate(_MON2)}
LL = [[_I2[tuple(a + b for a, b in zip(_MON1[i], _MON1[j]))] for j in range(4)] for i in range(4)]       # linear x linear -> quadratic slot
QL = [[_I3[tuple(a + b for a, b in zip(_MON2[i], _MON1[j]))] for j in range(4)] for i in range(10)]     # quadratic x linear -> cubic slot


def _mul_ll(a, b):
    out = np.zeros(10)
    for i in range(4):
        for j in range(4):
            out[LL[i][j]] += a[i] * b[j]
    return out


def _mul_ql(a, b):
    out = np.zeros(20)
    for i in range(10):
        for j in range(4):
            out[QL[i][j]] += a[i] * b[j]
    return out


def null_basis_5x9(Q):
    """4 vectors spanning the null space of the 5 x 9 system: Gauss-Jordan with full pivoting (largest |entry| of the remaining rows,
    first in row-major order on ties); free column f gives the vector with 1 at f and -R[i][f] at pivot column i.  Returns None if rank < 5"""
    A = np.array(Q, dtype=np.float64)
    piv = []
    for r in range(5):
        sub = np.abs(A[r:, :])
        k = int(np.argmax(sub))
        pr, pc = r + k // 9, k % 9
        if not sub.flat[k] > 1e-14:
            return None
        A[[r, pr]] = A[[pr, r]]
        A[r] /= A[r, pc]
        for i in range(5):
            if i != r:
                A[i] -= A[i, pc] * A[r]
        piv.append(pc)
    free = [c for c in range(9) if c not in piv]
    basis = []
    for f in free:
        v = np.zeros(9)
        v[f] = 1.0
        for i, pc in enumerate(piv):
            v[pc] = -A[i, f]
        basis.append(v)
    return basis


def _solve_6x5(C, d):
    """consistent 6 x 5 system by Gauss-Jordan with full pivoting over all six rows (csrc/pose_fivept.h solve_yz: same pivot rule)"""
    G = np.concatenate([C, d[:, None]], axis=1).astype(np.float64)
    perm = list(range(5))
    for c in range(5):
        sub = np.abs(G[c:, c:5])
        k = int(np.argmax(sub))
        pr, pc = c + k // (5 - c), c + k % (5 - c)
        if not sub.flat[k] > 0.0:
            return None
        G[[c, pr]] = G[[pr, c]]
        if pc != c:
            G[:, [c, pc]] = G[:, [pc, c]]
            perm[c], perm[pc] = perm[pc], perm[c]
        G[c, c:] /= G[c, c]
        for i in range(6):
            if i != c and G[i, c] != 0.0:
                G[i, c:] -= G[i, c] * G[c, c:]
    u = np.zeros(5)
    for c in range(5):
        u[perm[c]] = G[c, 5]
    return u


def five_point(x0, x1):
    """x0, x1 [5, 2] normalised points -> up to 10 essential matrices (Frobenius norm 1, x1h^T E x0h = 0), in ascending order of the
    eigenvalue (= the coefficient x of the first null vector)"""
    Q = np.stack([np.array([b[0] * a[0], b[0] * a[1], b[0], b[1] * a[0], b[1] * a[1], b[1], a[0], a[1], 1.0]) for a, b in zip(x0, x1)])
    basis = null_basis_5x9(Q)
    if basis is None:
        return []
    X, Y, Z, W = basis
    E = [[np.array([X[3 * i + j], Y[3 * i + j], Z[3 * i + j], W[3 * i + j]]) for j in range(3)] for i in range(3)]      # linear polynomials
    d = lambda a, b, c, e: _mul_ll(a, e) - _mul_ll(b, c)          # 2 x 2 minor
    det = _mul_ql(d(E[1][1], E[1][2], E[2][1], E[2][2]), E[0][0]) - _mul_ql(d(E[1][0], E[1][2], E[2][0], E[2][2]), E[0][1]) \
        + _mul_ql(d(E[1][0], E[1][1], E[2][0], E[2][1]), E[0][2])
    EEt = [[sum(_mul_ll(E[i][k], E[j][k]) for k in range(3)) for j in range(3)] for i in range(3)]
    tr = EEt[0][0] + EEt[1][1] + EEt[2][2]
    A = np.zeros((10, 20))
    A[0] = det
    for i in range(3):
        for j in range(3):
            A[1 + 3 * i + j] = 2.0 * sum(_mul_ql(EEt[i][k], E[k][j]) for k in range(3)) - _mul_ql(tr, E[i][j])
    try:
        B = np.linalg.solve(A[:, :10], A[:, 10:])
    except np.linalg.LinAlgError:
        return []
    M = np.zeros((10, 10))
    M[0:6] = -B[0:6]                 # x . [x^2, xy, xz, y^2, yz, z^2] = the degree-3 monomials x^3, x^2 y, x^2 z, x y^2, xyz, x z^2
    M[6, 0] = M[7, 1] = M[8, 2] = M[9, 6] = 1.0
    w = np.linalg.eigvals(M)
    out = []
    for lam in sorted(float(v.real) for v in w if v.imag == 0.0 and np.isfinite(v.real)):
        # the eigenvector b = [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1] of M for x = lam: rows 6..9 of M are unit rows, so b6 = lam, b0 = lam^2,
        # b1 = lam b7, b2 = lam b8 (b9 = 1), and rows 0..5 of (M - lam I) b = 0 are six consistent equations in u = (y^2, yz, z^2, y, z)
        C = np.zeros((6, 5))
        d = np.zeros(6)
        S = M - lam * np.eye(10)
        for r in range(6):
            C[r] = [S[r, 3], S[r, 4], S[r, 5], M[r, 7] + lam * S[r, 1], M[r, 8] + lam * S[r, 2]]
            d[r] = -(lam * lam * S[r, 0] + lam * M[r, 6] + M[r, 9])
        u = _solve_6x5(C, d)
        if u is None:
            continue
        x, y, z = lam, u[3], u[4]
        if not (np.isfinite(y) and np.isfinite(z)):
            continue
        Es = (x * X + y * Y + z * Z + W).reshape(3, 3)
        nrm = np.linalg.norm(Es)
        if nrm > 0 and np.isfinite(nrm):
            out.append(Es / nrm)
    return out


MAGSAC_K = 3.64
QUALITY_SCALE = 4096.0


def _gamma_u_3_2(x):
    from scipy.special import erfc
    rx = np.sqrt(x)
    return 0.88622692545275801 * erfc(rx) + rx * np.exp(-x)


def magsac_weight_exact(r2, sigma_max2):
    """MAGSAC++ (Barath et al., CVPR 2020) sigma-marginalised weight of a squared residual, normalised to w(0) = 1: noise scale uniform
    on (0, sigma_max], residuals chi-distributed with 4 degrees of freedom, inlier of scale sigma while r < 3.64 sigma (csrc/pose.hip)"""
    r2 = np.asarray(r2, dtype=np.float64)
    gk = _gamma_u_3_2(0.5 * MAGSAC_K ** 2)
    w = (_gamma_u_3_2(0.5 * r2 / sigma_max2) - gk) / (0.88622692545275801 - gk)
    return np.where(r2 < MAGSAC_K ** 2 * sigma_max2, w, 0.0)


WLUT_N = 2048
_WLUT = None


def _wlut():
    """the table of csrc/pose.hip (magsac_weight_lut): w at s = r / (k sigma_max) = j / WLUT_N; the last entry (the cut-off) is exactly 0"""
    global _WLUT
    if _WLUT is None:
        sj = np.arange(WLUT_N + 1, dtype=np.float64) / WLUT_N
        _WLUT = magsac_weight_exact(sj * sj * MAGSAC_K * MAGSAC_K, 1.0)
        _WLUT[WLUT_N] = 0.0
    return _WLUT


def magsac_weight(r2, sigma_max2):
    """the weight the kernels use: magsac_weight_exact tabulated in s = r / (k sigma_max) (WLUT_N intervals, linear interpolation; the
    published implementation tabulates the incomplete gamma function too) - same table, same arithmetic as csrc/pose.hip"""
    T = _wlut()
    r2 = np.asarray(r2, dtype=np.float64)
    s2 = r2 * (1.0 / (MAGSAC_K * MAGSAC_K * sigma_max2))
    inside = s2 < 1.0
    u = np.sqrt(np.where(inside, s2, 0.0)) * float(WLUT_N)
    j = np.minimum(u.astype(np.int64), WLUT_N - 1)
    f = u - j
    w = T[j] + f * (T[j + 1] - T[j])
    return np.where(inside, w, 0.0)


def _weighted_essential(x0, x1, w):
    """weighted 8-point fit (weights w >= 0) with weighted Hartley conditioning, projected onto the essential manifold"""
    def cond(x):
        c = (w[:, None] * x).sum(0) / w.sum()
        s = np.sqrt(2.0) / max((w * np.sqrt(((x - c) ** 2).sum(1))).sum() / w.sum(), 1e-12)
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.]])
        return (x - c) * s, T
    a, T0 = cond(x0)
    b, T1 = cond(x1)
    A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                  a[:, 0], a[:, 1], np.ones(len(a))], axis=1)
    _, V = np.linalg.eigh((A * w[:, None]).T @ A)
    E = T1.T @ V[:, 0].reshape(3, 3) @ T0
    U, sv, Vt = np.linalg.svd(E)
    if not (sv[1] > 1e-12 * sv[0] and sv[0] > 0):
        return None
    return U @ np.diag([1., 1., 0.]) @ Vt


def estimate_pose(kpts0, kpts1, K0, K1, norm_thresh, conf=0.99999, method=None, mask=None, iterations=4096, seed=1, return_consensus=False,
                  scoring='magsac', sampler='5pt', adaptive=False, info=None):
    """Signature of eval/pose_estimation.py:92 -> None | (E, R, t, mask).  Threshold: norm_thresh pixels divided by the
    mean focal length, applied to the Sampson distance in normalised coordinates.  ``mask`` follows :113-114 literally: all True,
    only the consensus entries overwritten by the cheirality result; ``return_consensus`` appends the geometric mask
    (consensus AND in front of both cameras).  ``scoring``: 'magsac' = sigma-marginalised quality + IRLS refinement (MAGSAC++ as
    published), 'count' = inlier counting + refits on the consensus set.  ``sampler``: '5pt' = five-point minimal solver (up to 10 models
    per sample; the reference's minimum of 5 matches, eval/pose_estimation.py:93), '8pt' = the linear eight-point solver of round 2
    (needs 8).  CPU twin of csrc/pose.hip in every mode."""
    n = len(kpts0)
    if n < (5 if sampler == '5pt' else 8):
        return None
    mag = scoring == 'magsac'
    x0, x1 = normalise(kpts0, K0), normalise(kpts1, K1)
    K0a, K1a = np.asarray(K0, dtype=np.float64), np.asarray(K1, dtype=np.float64)
    thr = norm_thresh / ((K0a[0, 0] + K0a[1, 1] + K1a[0, 0] + K1a[1, 1]) / 4.0)

    def weights(E):
        r2 = sampson_sq(E, x0, x1)
        return magsac_weight(r2, thr * thr) if mag else (r2 < thr * thr).astype(np.float64)

    def quality(E):
        q = weights(E).sum()
        return int(np.floor(q * QUALITY_SCALE)) if mag else int(q)

    best, bestE = -1, None
    ns = 5 if sampler == '5pt' else 8
    H1 = 128                                             # csrc/pose.hip POSE_H1
    need = iterations
    for h in range(iterations):
        if adaptive and sampler == '5pt' and iterations > H1 and h == H1:
            # adaptive termination (csrc/pose.hip pose_need_kernel, restated): the best support of the first H1 samples gives the inlier
            # ratio w; the smallest k with (1 - w^5)^k <= 1e-5 by repeated IEEE-double multiplication (no log: the same integer on both sides)
            k = iterations
            if best > 0:
                w = min(1.0, best / (QUALITY_SCALE * float(n)) if mag else best / float(n))
                pfail = 1.0 - w * w * w * w * w
                q, k = 1.0, 0
                while k < iterations:
                    q = q * pfail
                    k += 1
                    if q <= 1e-5:
                        break
            need = max(H1, k)
        if h >= need:
            break
        ids = [sample_index(seed, h, k, n) for k in range(ns)]
        if len(set(ids)) < ns:
            continue
        cands = five_point(x0[ids], x1[ids]) if sampler == '5pt' else [essential_from(x0[ids], x1[ids])]
        for E in cands:                                  # first best: lowest hypothesis, then lowest candidate index
            cnt = quality(E)
            if cnt > best:
                best, bestE = cnt, E
    if info is not None:
        info['samples'] = need
    if bestE is None or best < ((ns / 2) * QUALITY_SCALE if mag else ns):
        return None
    for rnd in range(3 if n >= 8 else 0):                # (weighted) least-squares refits (need 8 points), kept while not worse
        w = weights(bestE)
        E2 = _weighted_essential(x0, x1, w)
        if E2 is None:
            break
        q2 = quality(E2)
        if q2 < best:
            break
        same = q2 == best
        bestE, best = E2, q2
        if same and rnd > 0:
            break
    inl = sampson_sq(bestE, x0, x1) < thr * thr
    R, t, mP = decompose_essential_mat(bestE, np.asarray(kpts0)[inl], np.asarray(kpts1)[inl], K0, K1)
    m = np.ones(n, dtype=bool)                           # eval/pose_estimation.py:113: `E_mask.ravel() >= 0` - every entry True
    m[np.nonzero(inl)[0]] = mP                           # :114
    if return_consensus:
        geo = np.zeros(n, dtype=bool)
        geo[np.nonzero(inl)[0]] = mP
        return bestE, R, t, m, geo
    return bestE, R, t, m


def synthetic_scene(n, outliers=0.3, noise=0.5, seed=0, angle_deg=12.0):
    """two views of random 3D points with a known relative pose; returns kpts0, kpts1 (pixels), K, R, t (unit), inlier truth"""
    g = np.random.default_rng(seed)
    K = np.array([[520., 0, 320.], [0, 520., 240.], [0, 0, 1.]])
    X = np.stack([g.uniform(-2, 2, n), g.uniform(-1.5, 1.5, n), g.uniform(4, 9, n)], 1)
    ax = g.normal(size=3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(angle_deg)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
    t = g.normal(size=3); t /= np.linalg.norm(t)
    x0 = (K @ X.T).T; x0 = x0[:, :2] / x0[:, 2:]
    Xc = (R @ X.T).T + 0.8 * t
    x1 = (K @ Xc.T).T; x1 = x1[:, :2] / x1[:, 2:]
    x0 += g.normal(0, noise, x0.shape); x1 += g.normal(0, noise, x1.shape)
    truth = np.ones(n, dtype=bool)
    bad = g.permutation(n)[:int(outliers * n)]
    x1[bad] = np.stack([g.uniform(0, 640, len(bad)), g.uniform(0, 480, len(bad))], 1)
    truth[bad] = False
    return x0.astype(np.float32), x1.astype(np.float32), K, R, t, truth
