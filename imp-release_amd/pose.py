"""Pose step of the iterative loops on the GPU: drop-in for ``eval.pose_estimation.estimate_pose``.

``estimate_pose(kpts0, kpts1, K0, K1, norm_thresh, conf, method, mask)`` has the reference's signature
(eval/pose_estimation.py:92) and return value ``None | (E, R, t, inlier_mask)``, so it can be handed to
``matching_iterative[_uncertainty](..., estimate_pose=pose.estimate_pose)`` or patched over the reference's function.
The work runs in csrc/pose.hip: thousands of seeded minimal samples (five-point solver: up to 10 essential matrices each) scored in
parallel by the sigma-marginalised MAGSAC++ quality, IRLS refits of the winner, the ``decompose_essential_mat`` cheirality vote (eval/pose_estimation.py:13-89) - tens of microseconds of GPU time instead of a
host-side OpenCV call that parks the GPU 7 times per pair.

Same algorithm FAMILY as the reference's ``cv2.findEssentialMat(USAC_MAGSAC)`` since round 3 - five-point minimal solver, MAGSAC++
sigma-marginalised quality and IRLS refinement, all restated from the publications - but NOT that implementation: it is a third-party
randomized solver (opencv-contrib-python 4.5.5.64) with its own sampler, termination rule and local optimisation, no golden vectors in
the reference, and ``cv2`` is absent from the build image - parity with it is unpinned and not claimed (``method`` and ``conf`` are
accepted and ignored; the number of samples is fixed by ``iterations``).  Pinned: the kernels against their CPU
twin ``oracle/pose_oracle.py``, the cheirality vote against the geometric definition, recovery of known poses on synthetic
two-view scenes (tests/test_gpu_pose.py, tests/test_pose.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def estimate_pose(kpts0, kpts1, K0, K1, norm_thresh, conf=0.99999, method=None, mask=None, iterations=None, seed=1,
                  device=None, stream=None, return_consensus=False, scoring='magsac', sampler='5pt', adaptive=None):
    """eval/pose_estimation.py:92-115 -> None | (E [3,3], R [3,3], t [3], mask [n] bool).

    ``mask`` has the reference's semantics (:113-114: ``mask = E_mask.ravel() >= 0`` is all True, then only the consensus entries
    are overwritten with the cheirality result): matches OUTSIDE the consensus stay True.  ``return_consensus=True`` appends the
    geometric mask (in the consensus of E AND in front of both cameras).  ``scoring='magsac'`` (default): hypotheses ranked by the
    sigma-marginalised quality of MAGSAC++ and the winner refined by IRLS with those weights (the published algorithm behind the
    reference's ``cv2.USAC_MAGSAC``; OpenCV's implementation itself stays unpinned); ``'count'``: plain inlier counting + consensus refits.
    ``sampler='5pt'`` (default): minimal samples of 5 matches through the five-point solver (Nister / Stewenius-Engels-Nister, up to
    10 models per sample; csrc/pose_fivept.h) - like the reference, fewer than 5 matches -> None (eval/pose_estimation.py:93);
    ``'8pt'``: the linear eight-point sampler of round 2 (needs 8 matches).
    ``adaptive`` (round 5; five-point sampler; default off, IMP_POSE_ADAPTIVE=1 turns the default on - see _lib.default_pose_flags for the measurement): ``iterations`` is the CAP - after 128 samples the best support fixes how many are
    drawn at all, the smallest k with (1 - w^5)^k <= 1 - 0.99999 (include/imp_hip.h IMP_POSE_ADAPTIVE): the termination rule of the
    reference's USAC call (eval/pose_estimation.py:96-105, ``prob=conf``) in place of a fixed budget; same seeded sample sequence."""
    import torch
    if adaptive is None:
        adaptive = (_lib.default_pose_flags() & 4) != 0
    k0 = np.ascontiguousarray(np.asarray(kpts0, dtype=np.float32))
    k1 = np.ascontiguousarray(np.asarray(kpts1, dtype=np.float32))
    n = k0.shape[0]
    if sampler not in ('5pt', '8pt'):
        raise ValueError("sampler must be '5pt' or '8pt'")
    if iterations is None:
        # minimal samples: a five-point sample is all-inlier with probability w^5 instead of w^8 (w = 0.4: 1 in 98 instead of 1 in 1526)
        # and yields up to 10 models, so a quarter of the eight-point budget is ample (tools/probe/pose_compare.py)
        iterations = 1024 if sampler == '5pt' else 4096
    if n < (5 if sampler == '5pt' else 8) or k1.shape[0] != n:
        return None
    if scoring not in ('magsac', 'count'):
        raise ValueError("scoring must be 'magsac' or 'count'")
    if mask is not None:
        raise NotImplementedError('an input mask is not supported (no caller in the reference passes one)')
    L = _lib.lib()
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    Ka = np.ascontiguousarray(np.asarray(K0, dtype=np.float64).reshape(3, 3))
    Kb = np.ascontiguousarray(np.asarray(K1, dtype=np.float64).reshape(3, 3))
    E, R, t = np.zeros(9), np.zeros(9), np.zeros(3)
    m = np.zeros(n, dtype=np.uint8)
    cons = np.zeros(n, dtype=np.uint8)
    ninl = C.c_int()
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    P = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    rc = L.imp_estimate_pose(P(k0), P(k1), n, P(Ka), P(Kb), C.c_double(float(norm_thresh)), int(iterations), C.c_uint(seed), dev,
                             P(E), P(R), P(t), P(m), P(cons), C.byref(ninl), (1 if scoring == 'magsac' else 0) | (2 if sampler == '8pt' else 0) | (4 if adaptive and sampler == '5pt' else 0), C.c_void_p(st))
    if rc == 1:
        return None
    if rc != 0:
        raise _lib.ImpError(rc, 'imp_estimate_pose failed')
    if return_consensus:
        return E.reshape(3, 3), R.reshape(3, 3), t, m.astype(bool), cons.astype(bool)
    return E.reshape(3, 3), R.reshape(3, 3), t, m.astype(bool)


def pose_stats(reset=False):
    """(calls, minimal samples drawn) of the pose step in this process so far (imp_pose_stats)"""
    L = _lib.lib()
    a, b = C.c_long(), C.c_long()
    L.imp_pose_stats(C.byref(a), C.byref(b), 1 if reset else 0)
    return a.value, b.value
