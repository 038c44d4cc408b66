"""Pair datasets in the reference's dump layout and a pinned-memory prefetcher.

The reference evaluates from ONE HDF5 file written by ``dump/dumper/base_dumper.py:78-111`` and read by
``components/readers.py:8-33``: groups ``K1 K2 R T e f img_path1 img_path2 desc1 desc2 kpt1 kpt2``, each holding one
dataset per pair named ``str(index)``; ``kpt*`` rows are ``(x, y, score)``, ``desc*`` rows are descriptors, both cut to
``num_kpt`` rows on read; ``t = T / |T|``.  The reference then decodes both JPEGs per pair only to read ``.shape``
(``readers.py:28-29``, ``eval/eval_imp.py:46-48``); here the image sizes travel as two integers.

Backends
  * :class:`H5PairStore`  - the reference file itself: through ``h5py`` where it is installed, else through the built-in decoder
    :mod:`imp_release_amd.h5lite` (pure Python: the file-format subset h5py's defaults write - ``backend='h5lite'`` forces it).
  * :class:`NpzPairStore` - a directory of ``pair_<index>.npz`` files with the same field names (+ ``size1``, ``size2`` =
    (H, W) of the two images), written by :func:`write_npz_store`; what the tests and hosts without ``h5py`` use.

:func:`feed_data` builds the matcher's per-pair ``data`` dict exactly as ``eval/eval_imp.py:50-80`` does (HWC image
shape quirk included: the loops read ``image.shape[2:4]`` of a ``[1, H, W, 3]`` tensor);
:class:`PinnedPrefetcher` reads ahead on a host thread into pinned buffers and uploads on its own stream, so that the
per-pair host->device copy (N x 259 floats per image) overlaps the matcher's kernels.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Iterable, Optional, Sequence

import numpy as np
import torch

FIELDS = ('K1', 'K2', 'R', 'T', 'e', 'f', 'kpt1', 'kpt2', 'desc1', 'desc2')


def _finish(rec: dict, index: int, num_kpt: Optional[int]) -> dict:
    t = np.asarray(rec['T'])
    t = t / np.sqrt((t ** 2).sum())                                    # components/readers.py:16-17 (dtype and shape as stored)
    n = num_kpt if num_kpt else None
    return {'index': index, 'K1': np.asarray(rec['K1']), 'K2': np.asarray(rec['K2']), 'R': np.asarray(rec['R']), 't': t,
            'x1': np.asarray(rec['kpt1'])[:n], 'x2': np.asarray(rec['kpt2'])[:n],
            'desc1': np.asarray(rec['desc1'])[:n], 'desc2': np.asarray(rec['desc2'])[:n],
            'e': np.asarray(rec['e']), 'f': np.asarray(rec['f']), 'r_gt': np.asarray(rec['R']), 't_gt': t,
            'size1': tuple(int(v) for v in rec['size1']), 'size2': tuple(int(v) for v in rec['size2'])}


class NpzPairStore:
    def __init__(self, directory: str, num_kpt: Optional[int] = None):
        self.dir, self.num_kpt = directory, num_kpt
        self.n = len([f for f in os.listdir(directory) if f.startswith('pair_') and f.endswith('.npz')])

    def __len__(self):
        return self.n

    def record(self, index: int) -> dict:
        with np.load(os.path.join(self.dir, f'pair_{index}.npz')) as z:
            return _finish({k: z[k] for k in z.files}, index, self.num_kpt)


class H5PairStore:
    """the reference's ``*.hdf5`` dump; ``image_sizes`` = callable index -> ((H1, W1), (H2, W2)) or a fixed pair (the file
    stores image PATHS, not sizes: pass the sizes, or a function that looks them up, instead of decoding the JPEGs)"""

    def __init__(self, path: str, num_kpt: Optional[int] = None, *, image_sizes, backend: str = 'auto'):
        if backend not in ('auto', 'h5py', 'h5lite'):
            raise ValueError("backend: 'auto', 'h5py' or 'h5lite'")
        h5 = None
        if backend in ('auto', 'h5py'):
            try:
                import h5py as h5
            except ImportError:
                if backend == 'h5py':
                    raise
        if h5 is None:
            from . import h5lite as h5                                   # no h5py on this host: the built-in decoder
        self.backend = h5.__name__.rsplit('.', 1)[-1]
        self.f = h5.File(path, 'r')
        self.num_kpt, self.image_sizes = num_kpt, image_sizes

    def __len__(self):
        return len(self.f['K1'])

    def record(self, index: int) -> dict:
        rec = {k: self.f[k][str(index)][()] for k in FIELDS}
        s1, s2 = self.image_sizes(index) if callable(self.image_sizes) else self.image_sizes
        rec['size1'], rec['size2'] = s1, s2
        out = _finish(rec, index, self.num_kpt)
        out['img_path1'] = self.f['img_path1'][str(index)][()][0].decode()
        out['img_path2'] = self.f['img_path2'][str(index)][()][0].decode()
        return out

    def close(self):
        self.f.close()


def write_npz_store(records: Iterable[dict], directory: str) -> int:
    """records: dicts with the reference's field names (K1 K2 R T e f kpt1 kpt2 desc1 desc2) + size1, size2"""
    os.makedirs(directory, exist_ok=True)
    n = 0
    for i, r in enumerate(records):
        np.savez(os.path.join(directory, f'pair_{i}.npz'), **{k: np.asarray(r[k]) for k in FIELDS + ('size1', 'size2')})
        n += 1
    return n


def convert_h5_to_npz(h5_path: str, directory: str, *, image_sizes, backend: str = 'auto') -> int:
    store = H5PairStore(h5_path, None, image_sizes=image_sizes, backend=backend)
    def gen():
        for i in range(len(store)):
            rec = {k: store.f[k][str(i)][()] for k in FIELDS}
            rec['size1'], rec['size2'] = image_sizes(i) if callable(image_sizes) else image_sizes
            yield rec
    n = write_npz_store(gen(), directory)
    store.close()
    return n


def feed_data(rec: dict, device, host_buffers: Optional[dict] = None, stream=None) -> dict:
    """the per-pair ``data`` dict of eval/eval_imp.py:50-80 (device tensors + the host-side fields the loops read).
    ``host_buffers``: pinned staging tensors to reuse (see PinnedPrefetcher); ``stream``: upload stream."""
    x0, x1 = rec['x1'], rec['x2']
    host = {'keypoints0': x0[:, :2], 'scores0': x0[:, 2], 'descriptors0': rec['desc1'],
            'keypoints1': x1[:, :2], 'scores1': x1[:, 2], 'descriptors1': rec['desc2']}
    out = {}
    dev = torch.device(device)
    for k, a in host.items():
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None]
        if dev.type == 'cuda':
            if host_buffers is not None:
                buf = host_buffers.get(k)
                if buf is None or buf.shape != t.shape:
                    buf = host_buffers[k] = torch.empty(t.shape, dtype=torch.float32).pin_memory()
                buf.copy_(t)
                t = buf
            if stream is not None:
                with torch.cuda.stream(stream):
                    out[k] = t.to(dev, non_blocking=True)
            else:
                out[k] = t.to(dev, non_blocking=True)
        else:
            out[k] = t
    # only .shape of the images is ever read; [1, H, W, 3] like the reference's HWC upload (no pixels needed)
    out['image0'] = torch.empty((1,) + tuple(rec['size1']) + (3,), device='meta')
    out['image1'] = torch.empty((1,) + tuple(rec['size2']) + (3,), device='meta')
    out.update({'K0': rec['K1'], 'K1': rec['K2'], 'T_0to1': np.hstack([rec['R'], rec['t'].reshape(3, 1)]),
                'pts0_cpu': np.ascontiguousarray(x0[:, :2]), 'pts1_cpu': np.ascontiguousarray(x1[:, :2]),
                'E': rec.get('e'), 'index': rec['index']})
    return out


class PinnedPrefetcher:
    """iterates ``feed_data`` dicts for ``indices`` of ``store``; a host thread reads ``depth`` pairs ahead, stages them in
    pinned memory and uploads them on a private stream; the consumer's stream waits for the upload event of the pair it
    receives.  Use as ``for data in PinnedPrefetcher(store, range(len(store)), 'cuda'): ...``"""

    def __init__(self, store, indices: Sequence[int], device, depth: int = 3):
        self.store, self.indices, self.device, self.depth = store, list(indices), torch.device(device), max(1, depth)
        self.cuda = self.device.type == 'cuda'
        if self.cuda and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())

    def __iter__(self):
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def producer():
            try:
                if self.cuda:
                    torch.cuda.set_device(self.device)
                stream = torch.cuda.Stream(device=self.device) if self.cuda else None
                ring = [dict() for _ in range(self.depth + 2)]          # pinned staging sets, reused round-robin
                ring_ev = [None] * len(ring)
                for n, i in enumerate(self.indices):
                    if stop.is_set():
                        return
                    slot = n % len(ring)
                    if ring_ev[slot] is not None:
                        ring_ev[slot].synchronize()                     # the previous upload from this staging set is done
                    d = feed_data(self.store.record(i), self.device, ring[slot] if self.cuda else None, stream)
                    ev = None
                    if self.cuda:
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        ring_ev[slot] = ev
                    q.put((d, ev))
                q.put((None, None))
            except BaseException as ex:
                q.put((ex, None))

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                d, ev = q.get()
                if d is None:
                    return
                if isinstance(d, BaseException):
                    raise d
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for v in d.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(torch.cuda.current_stream(self.device))
                yield d
        finally:
            stop.set()
