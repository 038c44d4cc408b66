"""CPU: the pose oracle (oracle/pose_oracle.py) - the cheirality vote of eval/pose_estimation.py:13-89 against the geometric
definition, and the seeded 8-point RANSAC twin of csrc/pose.hip on synthetic two-view scenes with known poses.
(cv2 is absent: OpenCV / MAGSAC parity is unpinned and not claimed; see the module docstring.)"""
import numpy as np
import pytest

from oracle import pose_oracle as po


def _ang_mat(R1, R2):
    return np.rad2deg(np.abs(np.arccos(np.clip((np.trace(R1.T @ R2) - 1) / 2, -1, 1))))


def _ang_vec(a, b):
    return np.rad2deg(np.arccos(np.clip(a @ b / np.linalg.norm(a) / np.linalg.norm(b), -1, 1)))


def _skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


@pytest.mark.parametrize('seed', range(6))
def test_cheirality_vote_picks_the_true_pose_of_an_exact_essential_matrix(seed):
    k0, k1, K, R, t, _ = po.synthetic_scene(150, outliers=0.0, noise=0.0, seed=seed, angle_deg=5 + 7 * seed)
    for sign in (1.0, -1.0):                       # E and -E describe the same geometry
        Rd, td, m = po.decompose_essential_mat(sign * _skew(t) @ R, k0, k1, K, K)
        assert _ang_mat(R, Rd) < 1e-4 and _ang_vec(t, td) < 1e-3
        assert m.all()                             # every point is in front of both cameras for the true pose


def test_cheirality_masks_follow_the_geometric_definition():
    k0, k1, K, R, t, _ = po.synthetic_scene(80, outliers=0.0, noise=0.0, seed=3)
    g = np.random.default_rng(0)
    k1 = k1.copy()
    k1[:10] = np.stack([g.uniform(0, 640, 10), g.uniform(0, 480, 10)], 1)       # 10 wrong correspondences
    Rd, td, m = po.decompose_essential_mat(_skew(t) @ R, k0, k1, K, K)
    assert _ang_mat(R, Rd) < 1e-4
    # independent check: triangulated depth in both cameras (linear triangulation done here with lstsq)
    x0, x1 = po.normalise(k0, K), po.normalise(k1, K)
    for i in range(80):
        A = np.stack([np.array([-1, 0, x0[i, 0]]), np.array([0, -1, x0[i, 1]]),
                      x1[i, 0] * Rd[2] - Rd[0], x1[i, 1] * Rd[2] - Rd[1]])
        bvec = -np.array([0, 0, x1[i, 0] * td[2] - td[0], x1[i, 1] * td[2] - td[1]])
        X = np.linalg.lstsq(A, bvec, rcond=None)[0]
        zc = (Rd @ X + td)[2]
        if abs(X[2]) > 1e-3 and abs(zc) > 1e-3 and i >= 10:
            assert m[i] == (X[2] > 0 and zc > 0), i


def test_decompose_returns_proper_rotations_and_unit_translation():
    g = np.random.default_rng(5)
    for _ in range(20):
        R1, R2, t = po.decompose_E(g.normal(size=(3, 3)))
        for R in (R1, R2):
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-10) and np.isclose(np.linalg.det(R), 1.0)
        assert np.isclose(np.linalg.norm(t), 1.0)


def test_magsac_weight_function():
    """MAGSAC++ weight: 1 at r = 0, monotone, 0 from r = 3.64 sigma_max on, and the closed form of the incomplete gamma function
    Gamma_u(3/2, x) = sqrt(pi)/2 erfc(sqrt x) + sqrt x e^-x against scipy"""
    from scipy.special import gammaincc, gamma
    x = np.linspace(0, 9, 50)
    assert np.allclose(po._gamma_u_3_2(x), gammaincc(1.5, x) * gamma(1.5), atol=1e-12)
    s2 = 0.002 ** 2
    r = np.linspace(0, 4.0, 200) * 0.002
    for fn in (po.magsac_weight_exact, po.magsac_weight):       # the closed form, and the table the kernels (and the twin) read
        w = fn(r ** 2, s2)
        assert abs(w[0] - 1.0) < 1e-12 and (np.diff(w) <= 1e-15).all() and (w[r >= 3.64 * 0.002] == 0).all() and w[r < 3.6 * 0.002].min() > 0
    rr = np.random.default_rng(0).uniform(0, 3.7, 20000) * 0.002
    assert np.abs(po.magsac_weight(rr ** 2, s2) - po.magsac_weight_exact(rr ** 2, s2)).max() < 1e-6


def test_five_point_solver_contains_the_true_essential_matrix():
    """the five-point minimal solver (oracle twin of csrc/pose_fivept.h): on exact correspondences one of its <= 10 solutions is the true
    E = [t]x R, every solution satisfies the five epipolar constraints and the essential-matrix constraints"""
    found = 0
    for seed in range(20):
        k0, k1, K, R, t, _ = po.synthetic_scene(5, outliers=0.0, noise=0.0, seed=seed, angle_deg=5 + 2 * seed)
        x0, x1 = po.normalise(k0, K), po.normalise(k1, K)
        Et = _skew(t) @ R
        Et /= np.linalg.norm(Et)
        sols = po.five_point(x0, x1)
        assert 1 <= len(sols) <= 10
        for E in sols:
            h0, h1 = np.c_[x0, np.ones(5)], np.c_[x1, np.ones(5)]
            assert np.abs(np.einsum('ni,ij,nj->n', h1, E, h0)).max() < 1e-9
            assert abs(np.linalg.det(E)) < 1e-9 and np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-8
        found += min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in sols) < 1e-4      # (keypoints are stored as fp32)
    assert found >= 18


def test_five_point_header_of_the_gpu_kernel_equals_the_twin(tmp_path):
    """csrc/pose_fivept.h is plain C++: built for the host it must return the twin's solutions, in the twin's order"""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'fivept_host')
    subprocess.run(['g++', '-O2', '-std=c++17', os.path.join(root, 'tools', 'probe', 'fivept_host.cpp'), '-o', exe], check=True)
    scenes, lines = [], []
    for s in range(24):
        k0, k1, K, *_ = po.synthetic_scene(5, outliers=0.0 if s < 12 else 0.4, noise=0.0 if s < 6 else 0.5, seed=s)
        x0, x1 = po.normalise(k0, K), po.normalise(k1, K)
        scenes.append((x0, x1))
        lines += ['%.17g %.17g %.17g %.17g' % (a[0], a[1], b[0], b[1]) for a, b in zip(x0, x1)]
    out = subprocess.run([exe], input='\n'.join(lines) + '\n', capture_output=True, text=True, check=True).stdout.split('\n')
    pos = 0
    for x0, x1 in scenes:
        n = int(out[pos]); pos += 1
        got = [np.array(out[pos + k].split(), dtype=float).reshape(3, 3) for k in range(n)]; pos += n
        ref = po.five_point(x0, x1)
        assert len(ref) == n
        for a, b in zip(got, ref):
            assert np.abs(a - b).max() < 1e-7


def test_five_point_root_finder_against_lapack_on_random_samples(tmp_path):
    """round 4: the header finds the real eigenvalues of the action matrix as the real roots of its characteristic polynomial
    (derivative cascade + a Newton correction on the Hessenberg matrix), the twin asks LAPACK for all eigenvalues.  600 minimal samples
    drawn from scenes with outliers and noise (the samples RANSAC really meets: most are not all-inlier): the same number of real
    solutions for every sample, each solution within 5e-6 (the shifted-QR iteration of rounds 2-3 had the same worst case, 6e-7, on
    this set: what is left is the conditioning of the sample, not the solver)"""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'fivept_host')
    subprocess.run(['g++', '-O2', '-std=c++17', os.path.join(root, 'tools', 'probe', 'fivept_host.cpp'), '-o', exe], check=True)
    g = np.random.default_rng(5)
    scenes, lines = [], []
    for s in range(600):
        k0, k1, K, *_ = po.synthetic_scene(40, outliers=0.4 if s % 3 else 0.0, noise=[0.0, 0.3, 1.0][s % 3], seed=1000 + s)
        ids = g.choice(40, 5, replace=False)
        x0, x1 = po.normalise(k0[ids], K), po.normalise(k1[ids], K)
        scenes.append((x0, x1))
        lines += ['%.17g %.17g %.17g %.17g' % (a[0], a[1], b[0], b[1]) for a, b in zip(x0, x1)]
    out = subprocess.run([exe], input='\n'.join(lines) + '\n', capture_output=True, text=True, check=True).stdout.split('\n')
    pos, total, worst = 0, 0, 0.0
    for x0, x1 in scenes:
        n = int(out[pos]); pos += 1
        got = [np.array(out[pos + k].split(), dtype=float).reshape(3, 3) for k in range(n)]; pos += n
        ref = po.five_point(x0, x1)
        assert len(ref) == n
        total += n
        for a, b in zip(got, ref):
            worst = max(worst, np.abs(a - b).max())
    assert total > 2000 and worst < 5e-6, (total, worst)


@pytest.mark.parametrize('sampler,scoring', [('5pt', 'magsac'), ('5pt', 'count'), ('8pt', 'magsac'), ('8pt', 'count')])
@pytest.mark.parametrize('seed,outliers', [(0, 0.2), (1, 0.3), (2, 0.35)])
def test_ransac_twin_recovers_a_known_pose(seed, outliers, scoring, sampler):
    k0, k1, K, R, t, truth = po.synthetic_scene(500, outliers=outliers, noise=0.3, seed=seed)
    r = po.estimate_pose(k0, k1, K, K, 1.0, iterations=256 if sampler == '5pt' else 2048, seed=7, return_consensus=True, scoring=scoring,
                         sampler=sampler)
    assert r is not None
    E, Re, te, mref, m = r
    assert (mref | ~m).all() and (mref & ~m).sum() > 0      # the reference's mask (eval/pose_estimation.py:113-114) keeps non-consensus matches True
    assert _ang_mat(R, Re) < 2.0 and _ang_vec(t, te) < 8.0, (_ang_mat(R, Re), _ang_vec(t, te))
    assert (m & ~truth).sum() <= 0.05 * m.sum()            # hardly any outlier is accepted
    assert m.sum() >= 0.6 * truth.sum()


def test_too_few_matches_give_none():
    k0, k1, K, *_ = po.synthetic_scene(7, seed=1)
    assert po.estimate_pose(k0, k1, K, K, 1.0, sampler='8pt') is None
    assert po.estimate_pose(k0[:4], k1[:4], K, K, 1.0) is None                  # eval/pose_estimation.py:93


def test_sampling_hash_is_the_documented_one():
    # pinned values of the hypothesis-sampling hash shared with csrc/pose.hip (pose_rand); the GPU side is checked through the
    # identical consensus in tests/test_gpu_pose.py
    assert [po.sample_index(1, h, k, 1000) for h in (0, 1, 77) for k in (0, 7)] == [766, 495, 980, 244, 392, 577]
    assert len({po.sample_index(3, 5, k, 10 ** 6) for k in range(8)}) == 8
