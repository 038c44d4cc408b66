"""Iterative match + pose loops: the build's counterpart of ``eval/matching.py``.

Same signatures, schedule and return tuples as the reference:

* ``matching_iterative``             eval/matching.py:16-123   (IMP)
* ``matching_iterative_uncertainty`` eval/matching.py:126-276  (EIMP: adaptive pooling + real ragged slicing)

The GPU work goes through the step API of :mod:`imp_release_amd.modules` (HIP kernels).  The pose step
of the reference (``cv2.findEssentialMat(USAC_MAGSAC)``, eval/pose_estimation.py:92-115) is a CPU
third-party RANSAC that is out of scope here: pass it in as ``estimate_pose`` with the
reference's keyword signature; with ``estimate_pose=None`` no pose is ever found, the loop never exits
early and all iterations run (this is how the golden fixtures were captured from the reference).

Differences from the reference that do not change results: one device->host copy per valid iteration
(indices + scores) instead of three (the reference also copies the whole score matrix and never uses
it, eval/matching.py:70), and the ragged slicing is a HIP row-gather on token-major data.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .dist import pack_matches, unpack_matches
from .modules import VALID_ITS, _token_major


def angle_error_mat(R1, R2):
    """rotation angle (degrees) between two rotation matrices (tools/utils.py:425-431 semantics)"""
    cos = (np.trace(np.dot(R1.T, R2)) - 1) / 2
    return np.rad2deg(np.abs(np.arccos(np.clip(cos, -1., 1.))))


def angle_error_vec(v1, v2):
    """angle (degrees) between two vectors"""
    n = np.linalg.norm(v1) * np.linalg.norm(v2)
    return np.rad2deg(np.arccos(np.clip(np.dot(v1, v2) / n, -1.0, 1.0)))


def _normalize(model, data):
    if 'norm_keypoint0' in data.keys() and 'norm_keypoint1' in data.keys():      # sic: eval/matching.py:20,131
        return data['norm_keypoints0'], data['norm_keypoints1']
    ctx = model._ensure_ctx()
    _, _, h0, w0 = data['image0'].shape
    _, _, h1, w1 = data['image1'].shape
    return (ctx.normalize_keypoints(data['keypoints0'], w0, h0), ctx.normalize_keypoints(data['keypoints1'], w1, h1))


def _loop(data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method, estimate_pose, uncertainty,
          with_uncertainty, trace=None):
    ctx = model._ensure_ctx(check=True)
    return _loop_body(ctx, data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method, estimate_pose,
                      uncertainty, with_uncertainty, trace)


def _loop_body(ctx, data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method, estimate_pose, uncertainty,
               with_uncertainty, trace):
    norm_kpts0, norm_kpts1 = _normalize(model, data)
    pts0_cpu, pts1_cpu = data['pts0_cpu'], data['pts1_cpu']
    K0, K1 = data.get('K0'), data.get('K1')
    # desc + enc fused into the encoder's last GEMM epilogue (eval/matching.py:47-50,158-160)
    desc0, desc1 = ctx.encode_keypoints(norm_kpts0, data['scores0'], norm_kpts1, data['scores1'],
                                        data['descriptors0'], data['descriptors1'])
    last_best_R = last_best_t = None
    sel_ids0 = sel_ids1 = None
    sel_host0 = sel_host1 = None
    pred_score = None
    for it in range(nI):
        if uncertainty:
            if sel_ids0 is not None:                                              # eval/matching.py:166-169
                desc0 = ctx.gather_rows(desc0, sel_ids0)
                pts0_cpu = pts0_cpu[sel_host0 if sel_host0 is not None else sel_ids0.cpu().numpy()]
                norm_kpts0 = norm_kpts0[:, sel_ids0, :]
            if sel_ids1 is not None:                                              # eval/matching.py:171-174
                desc1 = ctx.gather_rows(desc1, sel_ids1)
                pts1_cpu = pts1_cpu[sel_host1 if sel_host1 is not None else sel_ids1.cpu().numpy()]
                norm_kpts1 = norm_kpts1[:, sel_ids1, :]
            sel_ids0 = sel_ids1 = sel_host0 = sel_host1 = None
        B, n0, n1 = desc0.shape[0], desc0.shape[1], desc1.shape[1]
        for li in (2 * it, 2 * it + 1):
            desc0, desc1 = ctx.forward_layer(li, desc0, desc1, inplace=True)
            model._note_layer(li, B, n0, n1)
        if it not in VALID_ITS:
            continue
        dist = ctx.compute_distance(it, desc0, desc1)
        for attempt in range(3):
            try:
                pred_score = ctx.compute_score(dist, model._bin(None), model.sinkhorn_iterations, model.with_sinkhorn)
                indices0, indices1, mscores0, mscores1 = ctx.compute_matches(pred_score, match_ratio)
                # the one sync of this iteration: indices and scores of image 0 in a single device->host copy
                packed = pack_matches(indices0[:1], mscores0[:1]).cpu()
            except _lib.ResidentSinkhornTimeout:
                # the health word was already raised when compute_matches (or the copy's entry check) looked at it - e.g. an uneven
                # spread over the XCCs is flagged at kernel start: the same event as the one caught below, the same cure
                continue
            # the copy synchronised: a voided chip-resident Sinkhorn launch (include/imp_hip.h, IMP_E_RESIDENT) shows now; the
            # context has then already stepped down to a safer protocol and the score is simply computed again
            if ctx.resident_health(raise_on_timeout=False) is not False:
                break
        else:
            raise _lib.ResidentSinkhornTimeout(_lib.IMP_E_RESIDENT, 'the Sinkhorn score stayed void on every protocol')
        idx_h, ms_h = unpack_matches(packed, n0)
        indices0_cpu, mscores0_cpu = idx_h[0].numpy(), ms_h[0].numpy()
        if trace is not None:
            trace.append({'it': it, 'n0': n0, 'n1': n1, 'indices0': indices0_cpu.copy(), 'mscores0': mscores0_cpu.copy(),
                          'pts0': pts0_cpu.copy(), 'pts1': pts1_cpu.copy()})
        matched_ids0 = np.nonzero(indices0_cpu > -1)[0]
        if matched_ids0.shape[0] < min_kpts:                                      # eval/matching.py:63-66
            last_best_R = last_best_t = None
            continue
        matched_ids1 = indices0_cpu[matched_ids0]
        if matched_ids0.shape[0] == 0:
            continue
        pred_matches = np.stack([matched_ids0, matched_ids1], axis=1)
        ret = None
        if estimate_pose is not None:
            ret = estimate_pose(kpts0=pts0_cpu[pred_matches[:, 0]], kpts1=pts1_cpu[pred_matches[:, 1]], K0=K0, K1=K1,
                                norm_thresh=error_th, method=method)
        if ret is not None:
            E, R, t, pose_inliers = ret
            inlier_ratio = np.sum(pose_inliers) / pred_matches.shape[0]
        else:
            R = t = None
            pose_inliers = np.zeros(pred_matches.shape[0], dtype=bool)
            inlier_ratio = 0
        if it >= 1:
            diff_R = angle_error_mat(last_best_R, R) if last_best_R is not None and R is not None else np.inf
            diff_t = angle_error_vec(last_best_t, t) if last_best_t is not None and t is not None else np.inf
        else:
            diff_R, diff_t = np.inf, np.inf
        pose_diff = np.max([diff_R, diff_t])
        last_best_R, last_best_t = R, t
        if uncertainty:                                                           # eval/matching.py:243-257
            mscore_th = 0.2 * inlier_ratio if (with_uncertainty and inlier_ratio != 0) else 0.2
            if hasattr(model, 'pool_host'):      # ids on the host from the same copy as the counts (no extra syncs)
                sel_ids0, sel_ids1, sel_host0, sel_host1 = model.pool_host(pred_score, mscore_th=mscore_th,
                                                                           uncertainty_ratio=1.0)
            else:
                sel_ids0, sel_ids1 = model.pool(pred_score=pred_score, prob00=model.self_prob0, prob01=model.cross_prob0,
                                                prob11=model.self_prob1, prob10=model.cross_prob1, mscore_th=mscore_th,
                                                uncertainty_ratio=1.0)
        if 'pose' in stop_criteria.keys() and pose_diff <= stop_criteria['pose']:  # eval/matching.py:110-117
            output_indice0 = np.zeros_like(indices0_cpu) - 1
            output_indice0[pred_matches[pose_inliers, 0]] = pred_matches[pose_inliers, 1]
            return pts0_cpu, pts1_cpu, norm_kpts0, norm_kpts1, output_indice0, mscores0_cpu, R, t, it + 1
    indices0, indices1, mscores0, mscores1 = ctx.compute_matches(pred_score, 0.2)  # eval/matching.py:119,271
    return (pts0_cpu, pts1_cpu, norm_kpts0, norm_kpts1, indices0[0].cpu().numpy(), mscores0[0].cpu().numpy(),
            None, None, nI)


def matching_iterative(data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method=None,
                       estimate_pose=None, trace=None):
    """eval/matching.py:16-123 -> (indices0, mscores0, R, t, n_iterations)"""
    r = _loop(data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method, estimate_pose, False, False,
              trace)
    return r[4], r[5], r[6], r[7], r[8]


def matching_iterative_uncertainty(data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method=None,
                                   with_uncertainty=False, estimate_pose=None, trace=None):
    """eval/matching.py:126-276 -> (pts0, pts1, norm_kpts0, norm_kpts1, indices0, mscores0, R, t, n_iterations)"""
    r = _loop(data, model, nI, match_ratio, min_kpts, error_th, stop_criteria, method, estimate_pose, True,
              with_uncertainty, trace)
    return (r[0], r[1], r[2][0].cpu().numpy(), r[3][0].cpu().numpy(), r[4], r[5], r[6], r[7], r[8])
