"""Batch-sharded evaluation of the iterative matchers over a list of independent image pairs
(BASELINE config 5 shape; the per-pair loop of eval/eval_imp.py:35-227 without the dataset / OpenCV parts).

Pairs are partitioned over ranks with :func:`imp_release_amd.dist.shard_range` (no data-path collective); each rank
runs ``matching_iterative`` (IMP) or ``matching_iterative_uncertainty`` (EIMP) on its pairs, and ONE all-gather at the
end collects a fixed-size summary row per pair: (n_iterations, n_matches, mean match score, n_kept0, n_kept1).
``estimate_pose`` is injected (the reference's cv2 MAGSAC step is out of scope); ``None`` = no early exit.
"""
from __future__ import annotations

from typing import Callable, Optional

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


def run_pairs_sharded(model, pair_provider: Callable[[int], dict], n_pairs: int, eimp: bool = False, nI: int = 15,
                      match_ratio: float = 0.1, min_kpts: int = 25, error_th: float = 1.0,
                      stop_criteria: Optional[dict] = None, estimate_pose=None, group=None) -> np.ndarray:
    """-> [n_pairs, 5] summary table, identical on every rank.  ``pair_provider(pair_id)`` returns the reference's
    per-pair ``data`` dict (GPU tensors + pts*_cpu / K*), exactly what eval/matching.py consumes."""
    stop_criteria = {'pose': 1.5} if stop_criteria is None else stop_criteria
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if ddp else 0
    world = dist.get_world_size(group) if ddp else 1
    s, e = shard_range(n_pairs, rank, world)
    rows = np.zeros((e - s, len(SUMMARY_COLUMNS)), dtype=np.float64)
    loop = matching.matching_iterative_uncertainty if eimp else matching.matching_iterative
    with torch.no_grad():
        for i, pid in enumerate(range(s, e)):
            out = loop(pair_provider(pid), model, nI, match_ratio, min_kpts, error_th, stop_criteria,
                       estimate_pose=estimate_pose)
            rows[i] = summarize(out, eimp)
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
