"""GPU: the pose step (csrc/pose.hip through imp_release_amd.pose.estimate_pose, reference signature
eval/pose_estimation.py:92) against its CPU twin oracle/pose_oracle.py, on synthetic two-view scenes with known poses, and
plugged into the iterative loops.  MAGSAC parity is unpinned (cv2 absent; stated in DESIGN.md)."""
import numpy as np
import pytest
import torch

from helpers import build_case, eval_config, load_golden, make_hip_model
from imp_release_amd import matching as hip_matching, pose as hip_pose, synthetic
from oracle import pose_oracle as po

pytestmark = pytest.mark.gpu


def _ang_mat(R1, R2):
    return np.rad2deg(np.abs(np.arccos(np.clip((np.trace(R1.T @ R2) - 1) / 2, -1, 1))))


def _ang_vec(a, b):
    return np.rad2deg(np.arccos(np.clip(a @ b / np.linalg.norm(a) / np.linalg.norm(b), -1, 1)))


@pytest.mark.parametrize('sampler,scoring', [('5pt', 'magsac'), ('5pt', 'count'), ('8pt', 'magsac'), ('8pt', 'count')])
@pytest.mark.parametrize('n,outliers,noise,seed', [(400, 0.3, 0.3, 0), (1500, 0.4, 0.5, 1), (60, 0.2, 0.2, 2), (3000, 0.5, 0.4, 3),
                                                   (9, 0.0, 0.0, 4), (6, 0.0, 0.1, 5), (5000, 0.35, 0.3, 6)])
def test_gpu_pose_equals_its_cpu_twin(n, outliers, noise, seed, scoring, sampler):
    """both samplers (five-point minimal solver, default; linear eight-point) and both rankings (sigma-marginalised MAGSAC++ quality with
    IRLS refinement, default; plain inlier counting) against the numpy twin: same samples, same algebra, same consensus
    (n = 5000: beyond the 4096 correspondences the refinement kernel keeps in registers)"""
    k0, k1, K, R, t, truth = po.synthetic_scene(n, outliers=outliers, noise=noise, seed=seed)
    its = 128 if sampler == '5pt' else 512
    g = hip_pose.estimate_pose(k0, k1, K, K, 1.0, iterations=its, seed=11, return_consensus=True, scoring=scoring, sampler=sampler)
    c = po.estimate_pose(k0, k1, K, K, 1.0, iterations=its, seed=11, return_consensus=True, scoring=scoring, sampler=sampler)
    assert (g is None) == (c is None)
    if g is None:
        return
    Eg, Rg, tg, mrg, mg = g
    Ec, Rc, tc, mrc, mc = c
    assert (mrg != mrc).sum() <= max(1, n // 500) and (mrg | ~mg).all()       # the reference-semantics mask: consensus outsiders stay True
    # same hypotheses, same algebra (fp64 both): the same consensus up to points sitting exactly on the threshold
    assert (mg != mc).sum() <= max(1, n // 500), (mg != mc).sum()
    assert _ang_mat(Rg, Rc) < 1e-3 and _ang_vec(tg, tc) < 1e-2
    s = np.sign((Eg * Ec).sum())
    assert np.abs(Eg - s * Ec).max() < 1e-6
    assert np.allclose(Rg @ Rg.T, np.eye(3), atol=1e-9) and np.isclose(np.linalg.det(Rg), 1.0) and np.isclose(np.linalg.norm(tg), 1.0)


@pytest.mark.parametrize('n,outliers,noise,seed', [(500, 0.2, 0.3, 7), (900, 0.45, 0.4, 8), (300, 0.7, 0.5, 9)])
def test_adaptive_termination_equals_the_cpu_twin(n, outliers, noise, seed):
    """round 5: adaptive termination - after 128 samples the best support fixes how many of the seeded samples are drawn
    (smallest k with (1 - w^5)^k <= 1e-5).  The bound is computed on the device by repeated multiplication in IEEE doubles, the numpy twin
    restates it: the SAME number of samples on both sides, hence the same models, consensus and pose; fewer samples than the cap on the
    clean scenes, the whole cap on the 70 %-outlier one; and the result equals a FIXED budget of exactly that many samples"""
    k0, k1, K, R, t, truth = po.synthetic_scene(n, outliers=outliers, noise=noise, seed=seed)
    cap = 512
    hip_pose.pose_stats(reset=True)
    g = hip_pose.estimate_pose(k0, k1, K, K, 1.0, iterations=cap, seed=11, return_consensus=True, adaptive=True)
    calls, drawn = hip_pose.pose_stats()
    info = {}
    c = po.estimate_pose(k0, k1, K, K, 1.0, iterations=cap, seed=11, return_consensus=True, adaptive=True, info=info)
    assert calls == 1 and drawn == info['samples'], (drawn, info)
    assert 128 <= drawn <= cap and (drawn < cap) == (outliers < 0.6), drawn
    assert (g is None) == (c is None) and g is not None
    Eg, Rg, tg, mrg, mg = g
    Ec, Rc, tc, mrc, mc = c
    assert (mg != mc).sum() <= max(1, n // 500), (mg != mc).sum()
    assert _ang_mat(Rg, Rc) < 1e-3 and _ang_vec(tg, tc) < 1e-2
    fixed = hip_pose.estimate_pose(k0, k1, K, K, 1.0, iterations=drawn, seed=11, return_consensus=True, adaptive=False)
    assert np.array_equal(fixed[0], Eg) and np.array_equal(fixed[4], mg)


@pytest.mark.parametrize('seed', range(4))
def test_gpu_pose_recovers_known_poses(seed):
    k0, k1, K, R, t, truth = po.synthetic_scene(1200, outliers=0.35, noise=0.3, seed=20 + seed, angle_deg=8 + 5 * seed)
    E, Re, te, mref, m = hip_pose.estimate_pose(k0, k1, K, K, 1.0, return_consensus=True)
    assert _ang_mat(R, Re) < 1.5 and _ang_vec(t, te) < 6.0, (_ang_mat(R, Re), _ang_vec(t, te))
    assert (m & ~truth).sum() <= 0.05 * m.sum() and m.sum() >= 0.6 * truth.sum()
    again = hip_pose.estimate_pose(k0, k1, K, K, 1.0, return_consensus=True)
    assert np.array_equal(again[4], m) and np.array_equal(again[3], mref) and np.array_equal(again[1], Re)          # deterministic (seeded)


def test_no_pose_cases():
    k0, k1, K, *_ = po.synthetic_scene(7, seed=1)
    assert hip_pose.estimate_pose(k0, k1, K, K, 1.0, sampler='8pt') is None
    assert hip_pose.estimate_pose(k0[:4], k1[:4], K, K, 1.0) is None            # eval/pose_estimation.py:93: fewer than 5 matches
    g = np.random.default_rng(0)
    k0 = g.uniform(0, 640, (50, 2)).astype(np.float32)
    k1 = g.uniform(0, 480, (50, 2)).astype(np.float32)                            # pure noise: whatever comes out is well formed
    r = hip_pose.estimate_pose(k0, k1, np.eye(3) * 500, np.eye(3) * 500, 0.5)
    assert r is None or (r[3].dtype == bool and r[3].shape == (50,))


def test_gpu_pose_drives_the_eimp_loop():
    """the loop with the GPU pose step in its estimate_pose slot (what eval/matching.py does with cv2): runs to an exit or to
    the end, returns well-formed outputs, and the early-exit indices are the inliers of the last pose"""
    spec, z = load_golden('eimp_loop_sliced_n1024')
    cfg, sd, data = build_case(spec, 'cuda')
    m = make_hip_model(spec, cfg, sd)
    d = dict(data)
    d['pts0_cpu'] = data['keypoints0'][0].cpu().numpy(); d['pts1_cpu'] = data['keypoints1'][0].cpu().numpy()
    d['K0'] = d['K1'] = np.array([[520., 0, 320.], [0, 520., 240.], [0, 0, 1.]]); d['T_0to1'] = np.eye(4)
    calls = []

    def gpu_pose(**kw):
        r = hip_pose.estimate_pose(**{k: v for k, v in kw.items() if k != 'method'})
        calls.append(None if r is None else int(r[3].sum()))
        return r

    with torch.no_grad():
        out = hip_matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5}, method=38, with_uncertainty=True,
                                                          estimate_pose=gpu_pose)
    p0, p1, nk0, nk1, i0, ms0, R, t, nit = out
    assert len(calls) >= 1 and i0.shape[0] == p0.shape[0] and ((i0 >= -1) & (i0 < p1.shape[0])).all()
    if R is not None:
        assert nit < 15 and int((i0 >= 0).sum()) == calls[-1] and np.isclose(np.linalg.det(R), 1.0)
    else:
        assert nit == 15
