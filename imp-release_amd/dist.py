"""Multi-GPU partitioning of image pairs.

Image pairs are independent (every op is per-sample, InstanceNorm is per-sample, the model caches are
per-instance), so the batch / pair list is sharded over ranks with NO data-path collective; weights are
replicated.  The only exchange is one all-gather of the per-pair results (indices0 int64 + mscores0 f32 =
12 bytes per keypoint), issued through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  One process per GPU.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """contiguous balanced block [start, stop) of ``n_items`` for ``rank`` (first n % world ranks get one more)"""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def pack_matches(indices0: torch.Tensor, mscores0: torch.Tensor) -> torch.Tensor:
    """[b, N] int64 + [b, N] float32 -> [b, N*12] uint8 (one buffer = one collective)"""
    b, n = indices0.shape
    a = indices0.contiguous().view(torch.uint8).reshape(b, n * 8)
    m = mscores0.contiguous().view(torch.uint8).reshape(b, n * 4)
    return torch.cat([a, m], dim=1)


def unpack_matches(buf: torch.Tensor, n: int):
    b = buf.shape[0]
    # fresh buffers with row pitches n*8 / n*4 (a one-row slice keeps the 12n pitch of `buf`, which need not be 8-aligned)
    a = torch.empty((b, n * 8), dtype=torch.uint8, device=buf.device).copy_(buf[:, :n * 8])
    m = torch.empty((b, n * 4), dtype=torch.uint8, device=buf.device).copy_(buf[:, n * 8:])
    return a.view(torch.int64).reshape(b, n), m.view(torch.float32).reshape(b, n)


def all_gather_matches(indices0: torch.Tensor, mscores0: torch.Tensor, n_total: int, group=None):
    """Every rank contributes the results of its shard_range() block; returns the full [n_total, N] tensors on
    every rank.  Blocks are padded to ceil(n_total / world) rows so that ONE equal-size all-gather suffices."""
    # IMP_FORCE_COLLECTIVES=1: go through the collective even on one rank (single-GPU check of the exchange lane)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and
                                                               not os.environ.get('IMP_FORCE_COLLECTIVES')):
        return indices0, mscores0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = indices0.shape[1]
    per = -(-n_total // world)
    local = pack_matches(indices0, mscores0)
    pad = torch.zeros(per, n * 12, dtype=torch.uint8, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty(world * per, n * 12, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    rows = []
    for r in range(world):
        s, e = shard_range(n_total, r, world)
        rows.append(out[r * per: r * per + (e - s)])
    return unpack_matches(torch.cat(rows, dim=0), n)


def all_gather_matches_steps(steps, n_total: int, group=None):
    """the results of SEVERAL batch-steps in ONE collective: ``steps`` = [(indices0 [b, N], mscores0 [b, N]), ...] of this rank's block ->
    [(indices0 [n_total, N], mscores0 [n_total, N]), ...] on every rank.  Every collective couples the ranks (it completes when the slowest has
    joined) and, on the GPU, its kernel holds compute units while it waits for a late peer - beside launches that need the whole chip
    (the chip-resident Sinkhorn, the fused layer launch) that is a stall for the fast rank; exchanging every K steps pays it once per K."""
    if not steps:
        return []
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and
                                                               not os.environ.get('IMP_FORCE_COLLECTIVES')):
        return list(steps)
    world = dist.get_world_size(group)
    n = steps[0][0].shape[1]
    per, k = -(-n_total // world), len(steps)
    pad = torch.zeros(k, per, n * 12, dtype=torch.uint8, device=steps[0][0].device)
    for j, (i0, m0) in enumerate(steps):
        pad[j, :i0.shape[0]] = pack_matches(i0, m0)
    out = torch.empty(world, k, per, n * 12, dtype=torch.uint8, device=pad.device)
    dist.all_gather_into_tensor(out.view(world * k, per, n * 12), pad, group=group)
    res = []
    for j in range(k):
        rows = []
        for r in range(world):
            s, e = shard_range(n_total, r, world)
            rows.append(out[r, j, :e - s])
        res.append(unpack_matches(torch.cat(rows, dim=0), n))
    return res


_QUEUE_SEQ = {}      # ranks of a group -> queues constructed for it so far


class DynamicPairQueue:
    """Rank-level DYNAMIC schedule for work of unpredictable cost ("dynamic work-stealing or longest-first" - the
    iterative loops leave after 6 ... 15 iterations, which no size-based static split can see): one shared counter, every rank pulls the
    next ``chunk`` item ids when it runs dry - ``next()`` -> list of ids (empty = done).  The counter lives in the job's rendezvous store
    (the TCPStore every torch.distributed job already has: ``store.add`` is an atomic fetch-and-add served by rank 0's store thread; no
    collective, nothing on the GPU), so a pull is one small TCP round trip (~0.1 ms) per chunk.  Without an initialised process group it
    is a local counter.  Every rank of a group must construct that group's queues in the same order (the key holds the group's ranks and a per-group
    sequence number)."""

    def __init__(self, n_items: int, chunk: int = 1, group=None):
        import threading
        self.n, self.chunk = int(n_items), max(1, int(chunk))
        self._lock = threading.Lock()
        self._local = 0
        self._store = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            from torch.distributed import distributed_c10d as c10d
            self._store = c10d._get_default_store()
            # the counter's key: the ranks of the group (two subgroups scheduling at the same time must not share a counter) and a per-group sequence number
            # (every rank of a group constructs that group's queues in the same order; a rank outside the group never touches the key)
            ranks = tuple(dist.get_process_group_ranks(group)) if group is not None else tuple(range(dist.get_world_size()))
            _QUEUE_SEQ[ranks] = _QUEUE_SEQ.get(ranks, 0) + 1
            import zlib
            self._key = f'imp_pair_queue/{zlib.crc32(repr(ranks).encode()):08x}/{len(ranks)}/{_QUEUE_SEQ[ranks]}'
        else:
            self._key = None
        self.pulls = 0

    def next(self):
        with self._lock:
            if self._store is not None:
                end = int(self._store.add(self._key, self.chunk))
                start = end - self.chunk
            else:
                start = self._local
                self._local += self.chunk
            self.pulls += 1
        return list(range(min(start, self.n), min(start + self.chunk, self.n)))


def lpt_assignment(costs, world: int):
    """Rank-level schedule for pairs of UNEQUAL cost (the iterative loops: cost grows with the keypoint counts, shrinks with pruning and
    early exit - SURVEY.md section 8(e) asks for "dynamic work-stealing or longest-first"): longest processing time first - pairs in
    order of decreasing cost, each to the rank with the least work so far (ties: lower pair id, lower rank).  Deterministic, computed
    identically on every rank from the same costs, no communication.  -> [[pair ids of rank 0], [rank 1], ...], each list ascending."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(x) for x in out]


def gpu_numa_cpus(local_rank: int, sysfs: str = '/sys'):
    """CPUs of the NUMA node the GPU `local_rank` hangs off (its PCI device's ``numa_node``), or None when the topology cannot be read
    (no such file, node -1: single-node hosts).  Looks the device up by PCI address through ``torch.cuda.get_device_properties`` when a
    GPU is visible, else by position among /sys/class/drm/card*/device entries that have a ``numa_node``."""
    import glob
    node = None
    try:
        if torch.cuda.is_available() and local_rank < torch.cuda.device_count():
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f'{getattr(pr, "pci_domain_id", 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
            with open(os.path.join(sysfs, 'bus', 'pci', 'devices', bdf, 'numa_node')) as f:
                node = int(f.read().strip())
    except Exception:
        node = None
    if node is None:
        cards = sorted(glob.glob(os.path.join(sysfs, 'class', 'drm', 'card[0-9]*', 'device', 'numa_node')))
        if local_rank < len(cards):
            try:
                with open(cards[local_rank]) as f:
                    node = int(f.read().strip())
            except Exception:
                node = None
    if node is None or node < 0:
        return None
    try:
        with open(os.path.join(sysfs, 'devices', 'system', 'node', f'node{node}', 'cpulist')) as f:
            txt = f.read().strip()
    except Exception:
        return None
    cpus = set()
    for part in txt.split(','):
        if '-' in part:
            a, b = part.split('-')
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return sorted(cpus) or None


def pin_to_gpu_numa(local_rank: int, sysfs: str = '/sys'):
    """One process per GPU: keep this rank's threads (the step workers, the exchange lane, the pose workers, the pinned-memory prefetcher)
    on the CPUs next to its GPU.  IMP_NUMA_AFFINITY=0 disables; a no-op when the topology cannot be read or the mask would be empty.
    -> the CPU list applied, or None."""
    if os.environ.get('IMP_NUMA_AFFINITY', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = gpu_numa_cpus(local_rank, sysfs)
    if not cpus:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    os.sched_setaffinity(0, allowed)
    return allowed
