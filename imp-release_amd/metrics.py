"""Metrics tail of the evaluation loop: host-side numpy, restated from the reference's definitions.

* ``compute_pose_error`` / ``angle_error_mat`` / ``angle_error_vec``  tools/utils.py:425-442
* ``pose_auc``  tools/utils.py:445-457  (exact area under the recall-vs-error curve up to each threshold)
* ``compute_epi_inlier``  components/utils/metrics.py:51-64  (symmetric point-to-epipolar-line distance)

Pinned against the imported reference by ``tools/make_golden.py`` -> ``tests/golden/metrics.npz``.
"""
from __future__ import annotations

import numpy as np

from .matching import angle_error_mat, angle_error_vec  # noqa: F401  (same definitions, re-exported)


def compute_pose_error(T_0to1, R, t):
    """-> (error_t, error_R) in degrees; the translation error folds the sign ambiguity of an essential matrix."""
    R_gt, t_gt = T_0to1[:3, :3], T_0to1[:3, 3]
    error_t = angle_error_vec(t, t_gt)
    error_t = np.minimum(error_t, 180 - error_t)
    return error_t, angle_error_mat(R, R_gt)


def pose_auc(errors, thresholds):
    """AUC of the cumulative recall curve, normalised by each threshold (README.md:149-154 reports @5/10/20 deg)."""
    e = np.sort(np.asarray(errors, dtype=np.float64))
    recall = (np.arange(len(e)) + 1) / len(e)
    e = np.concatenate([[0.], e])
    recall = np.concatenate([[0.], recall])
    out = []
    for t in thresholds:
        last = np.searchsorted(e, t)
        r = np.concatenate([recall[:last], [recall[last - 1]]])
        x = np.concatenate([e[:last], [t]])
        out.append(float(np.sum((x[1:] - x[:-1]) * (r[1:] + r[:-1]) / 2) / t))     # trapezoid rule
    return out


def compute_epi_inlier(x1, x2, E, inlier_th, return_error=False):
    """matches x1[i] <-> x2[i] (normalised coords): mean of the two point-to-epipolar-line distances < inlier_th"""
    x1_h = np.concatenate([x1, np.ones([x1.shape[0], 1])], -1)
    x2_h = np.concatenate([x2, np.ones([x2.shape[0], 1])], -1)
    l1 = x1_h @ E.T                      # epipolar lines of x1 in image 2
    l2 = x2_h @ E                        # epipolar lines of x2 in image 1
    norm = (1 / np.sqrt((l1[:, :2] ** 2).sum(1)) + 1 / np.sqrt((l2[:, :2] ** 2).sum(1))) / 2
    dis = np.abs((l1 * x2_h).sum(-1)) * norm
    mask = dis < inlier_th
    return (mask, dis) if return_error else mask
