"""Batch-sharded evaluation of the iterative matchers over a list of independent image pairs
(BASELINE config 5 shape; the per-pair loop of eval/eval_imp.py:35-227 without the dataset / OpenCV parts).

Pairs are partitioned over ranks with :func:`imp_release_amd.dist.shard_range` (no data-path collective); each rank
runs ``matching_iterative`` (IMP) or ``matching_iterative_uncertainty`` (EIMP) on its pairs, and ONE all-gather at the
end collects a fixed-size summary row per pair: (n_iterations, n_matches, mean match score, n_kept0, n_kept1).
``estimate_pose`` is injected (the reference's cv2 MAGSAC step is out of scope); ``None`` = no early exit.

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
from .dist import shard_range

SUMMARY_COLUMNS = ('n_iterations', 'n_matches', 'mean_mscore', 'n_kept0', 'n_kept1')


def summarize(out, eimp: bool) -> np.ndarray:
    if eimp:
        pts0, pts1, _, _, indices0, mscores0, _, _, n_iter = out
        k0, k1 = pts0.shape[0], pts1.shape[0]
    else:
        indices0, mscores0, _, _, n_iter = out
        k0 = k1 = -1
    valid = indices0 > -1
    mean = float(mscores0[valid].mean()) if valid.any() else 0.0
    return np.array([n_iter, int(valid.sum()), mean, k0, k1], dtype=np.float64)


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
                      replicas: Optional[Sequence] = None) -> np.ndarray:
    """-> [n_pairs, 5] summary table, identical on every rank.  ``pair_provider(pair_id)`` returns the reference's
    per-pair ``data`` dict (GPU tensors + pts*_cpu / K*), exactly what eval/matching.py consumes.
    ``workers`` > 1: that many pairs in flight on this rank (see module docstring); ``replicas`` may pass pre-built
    model instances (else they are created with :func:`replicate`)."""
    stop_criteria = {'pose': 1.5} if stop_criteria is None else stop_criteria
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if ddp else 0
    world = dist.get_world_size(group) if ddp else 1
    s, e = shard_range(n_pairs, rank, world)
    rows = np.zeros((e - s, len(SUMMARY_COLUMNS)), dtype=np.float64)
    loop = matching.matching_iterative_uncertainty if eimp else matching.matching_iterative

    def run_one(m, pid):
        out = loop(pair_provider(pid), m, nI, match_ratio, min_kpts, error_th, stop_criteria, estimate_pose=estimate_pose)
        return summarize(out, eimp)

    workers = max(1, min(int(workers), e - s))
    if workers == 1:
        with torch.no_grad():
            for i, pid in enumerate(range(s, e)):
                rows[i] = run_one(model, pid)
    else:
        models = list(replicas) if replicas is not None else replicate(model, workers)
        if len(models) < workers:
            raise ValueError(f'workers={workers} needs {workers} model replicas, got {len(models)}')
        device = model._device()
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        todo = iter(range(s, e))
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
                            pid = next(todo, None)
                        if pid is None:
                            return
                        if stream is not None:
                            with torch.cuda.stream(stream):
                                rows[pid - s] = run_one(m, pid)
                            stream.synchronize()
                        else:
                            rows[pid - s] = run_one(m, pid)
            except BaseException as ex:                  # surfaced on the calling thread
                errors.append(ex)

        threads = [threading.Thread(target=worker, args=(m,), daemon=True) for m in models[:workers]]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    if not ddp:
        return rows
    return gather_rows_across_ranks(rows, n_pairs, device=model._device() if hasattr(model, '_device') else 'cpu',
                                    group=group)


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
