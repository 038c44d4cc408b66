"""Batch-sharded evaluation of the iterative matchers over a list of independent image pairs
(BASELINE config 5 shape; the per-pair loop of eval/eval_imp.py:35-227 without the dataset / OpenCV parts).

Pairs are partitioned over ranks with :func:`imp_release_amd.dist.shard_range` (no data-path collective); each rank
runs ``matching_iterative`` (IMP) or ``matching_iterative_uncertainty`` (EIMP) on its pairs, and ONE all-gather at the
end collects a fixed-size summary row per pair (``SUMMARY_COLUMNS``: pose errors, precision, matching score - the metrics tail of
eval/eval_imp.py:112-141,190-225 through :mod:`imp_release_amd.metrics` - and the loop statistics); :func:`aggregate` turns the
table into the reference's report (AUC@5/10/20/50, precision, matching score).
``estimate_pose`` is injected (``imp_release_amd.pose.estimate_pose`` = the GPU pose step; the reference's cv2 MAGSAC call when
available); ``None`` = no pose, no early exit, infinite pose errors.

Pairs in flight: one pair at batch 1 cannot fill the GPU (every kernel of the loop is a few workgroups, the step is
bound by ~7 us of dependent-launch latency per kernel), and the reference's loop additionally parks the GPU during
every host-side pose estimate (eval/matching.py:63-70).  ``workers=K`` runs K pairs
concurrently: K replicas of the model (each owns its workspace and attention cache - the modules are stateful, like
the reference's), one host thread and one stream each; a pair's pose estimate overlaps the other pairs' kernels.
Per-pair results do not depend on K (every kernel has a fixed reduction order).
"""
from __future__ import annotations

import copy
import threading
from typing import Callable, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import matching
from .dist import lpt_assignment, shard_range

# one row per pair = what eval/eval_imp.py:112-141,190-225 accumulates for its report: pose errors (degrees; inf = no pose),
# precision and matching score of the final matches against the ground-truth essential matrix, plus the loop statistics
SUMMARY_COLUMNS = ('err_R', 'err_t', 'precision', 'matching_score', 'mean_mscore', 'n_iterations', 'n_matches', 'n_kept0', 'n_kept1')
AUC_THRESHOLDS = (5, 10, 20, 50)             # eval/eval_imp.py:36


def normalize_intrinsic(x, K):
    """components/utils/evaluation_utils.py:6-8"""
    K = np.asarray(K, dtype=np.float64)
    return (np.asarray(x, dtype=np.float64) - K[:2, 2]) / np.diag(K)[:2]


def summarize(out, eimp: bool, data: Optional[dict] = None, estimate_pose=None, error_th: float = 1.0) -> np.ndarray:
    """the metrics tail of eval/eval_imp.py for one pair (:112-141 iterative branch): final matches -> epipolar precision /
    matching score against the ground-truth E (``data['E']``, inlier_th 0.005 in intrinsics-normalised coordinates), pose error
    of the loop's pose - or, when the loop found none, of ``estimate_pose`` on the final matches - against ``data['T_0to1']``.
    Rows of pairs without ground truth carry NaN in the first four columns.  (For the EIMP loop the reference re-uses the
    loop's image-size-normalised keypoint tensor as if it were intrinsics-normalised, eval/eval_imp.py:95,127-128; this
    function normalises the loop's surviving pixel coordinates with K, which is what the formula expects.)"""
    from . import metrics
    if eimp:
        pts0, pts1, _, _, indices0, mscores0, pred_R, pred_t, n_iter = out
        k0, k1 = pts0.shape[0], pts1.shape[0]
    else:
        indices0, mscores0, pred_R, pred_t, n_iter = out
        k0 = k1 = -1
        pts0 = None if data is None else data.get('pts0_cpu')
        pts1 = None if data is None else data.get('pts1_cpu')
    valid = indices0 > -1
    mean = float(mscores0[valid].mean()) if valid.any() else 0.0
    err_R = err_t = precision = mscore = float('nan')
    if data is not None and pts0 is not None and 'K0' in data and 'T_0to1' in data:
        K0, K1 = np.asarray(data['K0'], dtype=np.float64), np.asarray(data['K1'], dtype=np.float64)
        mk0, mk1 = np.asarray(pts0)[valid], np.asarray(pts1)[indices0[valid]]
        if data.get('E') is not None:
            correct = metrics.compute_epi_inlier(normalize_intrinsic(mk0, K0), normalize_intrinsic(mk1, K1),
                                                 np.asarray(data['E'], dtype=np.float64), 0.005) if len(mk0) else np.zeros(0, dtype=bool)
            precision = float(np.mean(correct)) if len(correct) > 0 else 0.0
            mscore = float(np.sum(correct)) / len(pts0) if len(pts0) > 0 else 0.0
        R, t = pred_R, pred_t
        if R is None and estimate_pose is not None:
            ret = estimate_pose(kpts0=mk0, kpts1=mk1, K0=K0, K1=K1, norm_thresh=error_th)
            if ret is not None:
                R, t = ret[1], ret[2]
        if R is None:
            err_t, err_R = float('inf'), float('inf')
        else:
            err_t, err_R = (float(v) for v in metrics.compute_pose_error(np.asarray(data['T_0to1'], dtype=np.float64), R, t))
    return np.array([err_R, err_t, precision, mscore, mean, n_iter, int(valid.sum()), k0, k1], dtype=np.float64)


def aggregate(table: np.ndarray) -> dict:
    """the report of eval/eval_imp.py:213-227 from the gathered table: pose AUC@5/10/20/50 of max(err_R, err_t) (tools/utils.py:445-457),
    mean precision / matching score (percent), mean loop statistics"""
    from . import metrics
    col = {n: table[:, i] for i, n in enumerate(SUMMARY_COLUMNS)}
    out = {'pairs': int(table.shape[0])}
    have = ~np.isnan(col['err_R'])
    if have.any():
        pose_errors = np.maximum(col['err_R'][have], col['err_t'][have])
        for th, a in zip(AUC_THRESHOLDS, metrics.pose_auc(pose_errors, AUC_THRESHOLDS)):
            out[f'auc@{th}'] = round(100.0 * a, 2)
        out['precision'] = round(100.0 * float(np.nanmean(col['precision'][have])), 2)
        out['matching_score'] = round(100.0 * float(np.nanmean(col['matching_score'][have])), 2)
        out['pose_found'] = round(float(np.mean(np.isfinite(pose_errors))), 3)
    for n in ('mean_mscore', 'n_iterations', 'n_matches', 'n_kept0', 'n_kept1'):
        out[n] = round(float(col[n].mean()), 3)
    return out


def replicate(model, n: int) -> list:
    """[model, copy, copy, ...]: n independent instances with the same weights on the same device"""
    reps = [model]
    for _ in range(n - 1):
        m = type(model)(copy.deepcopy(model.config)).eval()
        m.load_state_dict(model.state_dict(), strict=True)
        reps.append(m.to(model._device()))
    return reps


def run_pairs_sharded(model, pair_provider: Callable[[int], dict], n_pairs: int, eimp: bool = False, nI: int = 15,
                      match_ratio: float = 0.1, min_kpts: int = 25, error_th: float = 1.0,
                      stop_criteria: Optional[dict] = None, estimate_pose=None, group=None, workers: int = 1,
                      replicas: Optional[Sequence] = None, lockstep: int = 1, schedule: str = 'block',
                      pair_cost: Optional[Callable[[int], float]] = None, with_uncertainty: Optional[bool] = None,
                      group_similar: int = 0) -> np.ndarray:
    """-> [n_pairs, len(SUMMARY_COLUMNS)] summary table, identical on every rank.  ``pair_provider(pair_id)`` returns the reference's
    per-pair ``data`` dict (GPU tensors + pts*_cpu / K*), exactly what eval/matching.py consumes.
    ``workers`` > 1: that many pairs in flight on this rank (see module docstring); ``replicas`` may pass pre-built
    model instances (else they are created with :func:`replicate`).
    ``lockstep`` > 1 (round 4): that many pairs advance TOGETHER through ``matching.matching_iterative_lockstep`` (IMP) /
    ``matching.matching_iterative_uncertainty_lockstep`` (EIMP: per-pair pooling inside the ragged batch) - one ragged batch, one kernel
    launch per layer for all of them, per-pair early exit (at most 4 pairs of ~2048 keypoints, 8 of <= 1024: the batch must fit the
    chip-resident Sinkhorn); with ``workers`` > 1 several such groups are in flight.  A pair's row is the row of the pair run alone, bit for
    bit (round 6: every reduction order is a function of the pair's own sizes - tests/test_gpu_batch_invariance.py, tests/test_gpu_hard_loops.py;
    tools/probe/eimp_lockstep_diff.py: 0 of 96 pairs of the harder set differ; rounds 4-5: 11 of 96 kept another keypoint set in a group).
    ``group_similar`` = W > 0 (with ``lockstep`` > 1 and ``pair_cost``): every window of W consecutive pairs of this rank is taken in
    descending cost order - what a loader with a look-ahead of W pairs can do - so that the pairs of a group have similar sizes (a group is
    padded to its largest pair and advances at that pair's pace); rows stay in pair-id order.
    ``with_uncertainty`` (EIMP): pool threshold 0.2 x the pose estimate's inlier ratio (eval/matching.py:243-247); default = ``eimp``,
    as eval/eval_imp.py:95-105 passes its one ``use_uncertainty`` switch to both.
    ``schedule``: how pairs map to ranks - 'block' (contiguous blocks, :func:`imp_release_amd.dist.shard_range`), 'lpt' (longest
    processing time first over ``pair_cost(pid)``: see :func:`imp_release_amd.dist.lpt_assignment`) or 'dynamic' (round 5: every rank pulls
    the next ``lockstep`` - or ``group_similar`` - pairs from one shared counter when it runs dry, :class:`imp_release_amd.dist.DynamicPairQueue`:
    balances what no static split can see, the exit iteration; the table is the same whatever rank evaluated a pair)."""
    stop_criteria = {'pose': 1.5} if stop_criteria is None else stop_criteria
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if ddp else 0
    world = dist.get_world_size(group) if ddp else 1
    queue = None
    if schedule == 'lpt':
        if pair_cost is None:
            raise ValueError("schedule='lpt' needs pair_cost(pair_id) -> relative cost (e.g. n0 * n1)")
        mine = lpt_assignment([pair_cost(i) for i in range(n_pairs)], world)[rank]
    elif schedule == 'block':
        mine = list(range(*shard_range(n_pairs, rank, world)))
    elif schedule == 'dynamic':
        # pairs are PULLED: one shared counter in the job's store, every rank takes the next chunk when it runs dry (dist.DynamicPairQueue) -
        # the schedule for work whose cost shows only while it runs (the loops' early exit: 6 ... 15 iterations)
        from .dist import DynamicPairQueue
        lk = max(1, int(lockstep))
        queue = DynamicPairQueue(n_pairs, chunk=max(lk, int(group_similar) if (group_similar and lk > 1) else lk), group=group)
        mine = []
    else:
        raise ValueError("schedule: 'block', 'lpt' or 'dynamic'")
    if group_similar and lockstep > 1 and queue is None:
        if pair_cost is None:
            raise ValueError('group_similar needs pair_cost(pair_id)')
        W = max(1, int(group_similar))
        mine = [i for a in range(0, len(mine), W) for i in sorted(mine[a:a + W], key=lambda i: (-pair_cost(i), i))]
    pos = {pid: i for i, pid in enumerate(mine)}

    class _Rows:                                  # rows[pid - s] of the block schedule, for any id list
        def __init__(self):
            self.a = np.zeros((len(mine), len(SUMMARY_COLUMNS)), dtype=np.float64)
            self.d = {}

        def __setitem__(self, k, v):
            if queue is not None:
                self.d[k] = np.asarray(v, dtype=np.float64)          # (dict stores are atomic: worker threads write disjoint keys)
            else:
                self.a[pos[k]] = v
    s = 0
    rows = _Rows()
    loop = matching.matching_iterative_uncertainty if eimp else matching.matching_iterative
    unc = dict(with_uncertainty=bool(eimp if with_uncertainty is None else with_uncertainty)) if eimp else {}
    lockstep = max(1, int(lockstep))
    group_loop = matching.matching_iterative_uncertainty_lockstep if eimp else matching.matching_iterative_lockstep
    # units of work: single pairs, or groups of `lockstep` pairs that advance together (pairs of similar cost side by side under 'lpt':
    # the list is ascending in pair id, the provider's order)
    units = [mine[a:a + lockstep] for a in range(0, len(mine), lockstep)]
    if queue is not None:
        def _pulled():
            while True:
                ids = queue.next()
                if not ids:
                    return
                if group_similar and lockstep > 1 and pair_cost is not None:
                    ids = sorted(ids, key=lambda i: (-pair_cost(i), i))
                for a in range(0, len(ids), lockstep):
                    yield ids[a:a + lockstep]
        units = _pulled()

    def run_unit(m, pids):
        from . import _lib
        if len(pids) == 1 and lockstep == 1:
            data = pair_provider(pids[0])
            try:
                out = loop(data, m, nI, match_ratio, min_kpts, error_th, stop_criteria, estimate_pose=estimate_pose, **unc)
            except _lib.ResidentSinkhornTimeout:      # a voided waiting launch outside the score step (a fused layer, an entry check): the pair once more, on the protocol the context stepped down to
                out = loop(data, m, nI, match_ratio, min_kpts, error_th, stop_criteria, estimate_pose=estimate_pose, **unc)
            rows[pids[0] - s] = summarize(out, eimp, data, estimate_pose, error_th)
            return
        datas = [pair_provider(pid) for pid in pids]
        try:
            outs = group_loop(datas, m, nI, match_ratio, min_kpts, error_th, stop_criteria, estimate_pose=estimate_pose, **unc)
        except _lib.ResidentSinkhornTimeout:          # a voided launch inside the pipelined group: once more, on the protocol the context stepped down to
            outs = group_loop(datas, m, nI, match_ratio, min_kpts, error_th, stop_criteria, estimate_pose=estimate_pose, **unc)
        for pid, data, out in zip(pids, datas, outs):
            rows[pid - s] = summarize(out, eimp, data, estimate_pose, error_th)

    workers = max(1, int(workers)) if queue is not None else max(1, min(int(workers), len(units)))
    if workers == 1:
        with torch.no_grad():
            for u in units:
                run_unit(model, u)
    else:
        models = list(replicas) if replicas is not None else replicate(model, workers)
        if len(models) < workers:
            raise ValueError(f'workers={workers} needs {workers} model replicas, got {len(models)}')
        device = model._device()
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        todo = iter(units)
        lock = threading.Lock()
        errors = []

        def worker(m):
            stream = torch.cuda.Stream(device=device) if device.type == 'cuda' else None
            try:
                if stream is not None:
                    torch.cuda.set_device(device)            # a new thread starts with device 0 current
                with torch.no_grad():
                    while not errors:
                        with lock:
                            unit = next(todo, None)
                        if unit is None:
                            return
                        if stream is not None:
                            with torch.cuda.stream(stream):
                                run_unit(m, unit)
                            stream.synchronize()
                        else:
                            run_unit(m, unit)
            except BaseException as ex:                  # surfaced on the calling thread
                errors.append(ex)
            finally:
                matching.release_thread_resources()      # this thread's pose workers / pinned staging die with it

        threads = [threading.Thread(target=worker, args=(m,), daemon=True) for m in models[:workers]]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    if queue is not None:
        mine = sorted(rows.d)
        rows.a = np.stack([rows.d[i] for i in mine]) if mine else np.zeros((0, len(SUMMARY_COLUMNS)), dtype=np.float64)
    if not ddp:
        return rows.a[np.argsort(np.asarray(mine, dtype=np.int64), kind='stable')] if len(mine) else rows.a       # rows in pair-id order
    return gather_rows_by_id(rows.a, mine, n_pairs, device=model._device() if hasattr(model, '_device') else 'cpu', group=group)


def gather_rows_by_id(rows: np.ndarray, ids, n_total: int, device='cpu', group=None) -> np.ndarray:
    """one equal-size all-gather of (pair id, row) blocks padded to the largest rank's count -> the full table in pair order (any
    assignment of pairs to ranks: contiguous blocks or dist.lpt_assignment)"""
    world, width = dist.get_world_size(group), rows.shape[1]
    cnt = torch.tensor([len(ids)], dtype=torch.int64, device=device)
    cnts = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    per = int(cnts.max().item())
    pad = torch.full((per, width + 1), -1.0, dtype=torch.float64, device=device)
    if len(ids):
        pad[:len(ids), 0] = torch.tensor(list(ids), dtype=torch.float64, device=device)
        pad[:len(ids), 1:] = torch.from_numpy(rows).to(device)
    out = torch.empty(world * per, width + 1, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.cpu().numpy()
    table = np.full((n_total, width), np.nan)
    valid = out[:, 0] >= 0
    got = out[valid, 0].astype(np.int64)
    # every pair exactly once: a schedule whose ranks disagree (two groups on one counter, queues built in different orders) skips or repeats pairs
    counts = np.bincount(got, minlength=n_total)
    if (counts != 1).any():
        raise RuntimeError(f'gather_rows_by_id: {int((counts == 0).sum())} of {n_total} pairs were evaluated by no rank, {int((counts > 1).sum())} by several - '
                           f'the ranks did not agree on the schedule')
    table[got] = out[valid, 1:]
    return table


def gather_rows_across_ranks(rows: np.ndarray, n_total: int, device='cpu', group=None) -> np.ndarray:
    """equal-size all-gather of the per-rank blocks (padded to ceil(n_total / world) rows)"""
    world, width = dist.get_world_size(group), rows.shape[1]
    per = -(-n_total // world)
    pad = torch.zeros(per, width, dtype=torch.float64, device=device)
    pad[:rows.shape[0]] = torch.from_numpy(rows).to(device)
    out = torch.empty(world * per, width, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, pad, group=group)
    blocks = []
    for r in range(world):
        s, e = shard_range(n_total, r, world)
        blocks.append(out[r * per: r * per + (e - s)])
    return torch.cat(blocks, 0).cpu().numpy()
