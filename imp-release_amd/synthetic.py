"""Deterministic synthetic weights and inputs for the IMP/EIMP matching hot path.

No pretrained weights or datasets exist in this environment, so parity and
benchmarks are pinned on *seeded synthetic* state_dicts and SuperPoint-like inputs.  Everything
here is numpy (PCG64) so that the very same tensors can be regenerated on the GPU box without
shipping 77 MB of weights or any reference file.

The key schema follows the reference's ``state_dict`` (probed from
``nets/gm.py:58-75``, ``nets/layers.py:59-90,100-107,139-149,182-198``):

* ``bin_score`` ()
* ``kenc.encoder.{0,3,6,9,12}.{weight,bias}``
* non-shared layer i : ``gnn.layers.i.attn.proj.{0,1,2}.*``, ``gnn.layers.i.attn.merge.*``,
  ``gnn.layers.i.mlp.{0,3}.*``
* shared layer i     : ``gnn.layers.i.proj.*``, ``gnn.layers.i.merge.*``, ``gnn.layers.i.mlp.{0,3}.*``
* ``final_proj.{0..n_layers-1}.*``
* with ``norm_fn='bn'``: ``...encoder.{1,4,7,10}`` / ``...mlp.1`` get
  ``weight,bias,running_mean,running_var,num_batches_tracked``.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

DEFAULT_CONFIG = {
    # mirrors nets/gm.py:30-44 (values only; the reference file is not imported here)
    'descriptor_dim': 256,
    'weights': 'indoor',
    'keypoint_encoder': [32, 64, 128, 256],
    'GNN_layers': ['self', 'cross'] * 9,
    'sinkhorn_iterations': 20,
    'match_threshold': 0.2,
    'with_pose': False,
    'n_layers': 9,
    'n_min_tokens': 256,
    'with_sinkhorn': True,
    'ac_fn': 'relu',
    'norm_fn': 'bn',
}


def sharing_pattern(n_gnn_layers: int, model: str):
    """Which GNN layers re-use the previous iteration's attention (nets/gms.py:17, nets/adgm.py:18)."""
    if model == 'GM':
        return [False] * n_gnn_layers
    pat = [False, False] * 2 + [False, False, True, True] * 21
    return pat[:n_gnn_layers]


def _rng_for(seed: int, key: str) -> np.random.Generator:
    # independent stream per tensor name: order of generation never matters
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def _conv(sd, seed, prefix, cout, cin, zero_bias=False, gain=1.0, bias_offset=0.0):
    bound = gain / np.sqrt(cin)
    w = _rng_for(seed, prefix + '.weight').uniform(-bound, bound, size=(cout, cin, 1))
    b = _rng_for(seed, prefix + '.bias').uniform(-bound, bound, size=(cout,))
    if zero_bias:
        b[:] = 0.0
    if bias_offset:
        # channels whose mean is far above their spread (alternating sign): the stress case for InstanceNorm statistics
        b = b + bias_offset * np.where(np.arange(cout) % 2 == 0, 1.0, -1.0)
    sd[prefix + '.weight'] = w.astype(np.float32)
    sd[prefix + '.bias'] = b.astype(np.float32)


def _bn(sd, seed, prefix, c):
    sd[prefix + '.weight'] = _rng_for(seed, prefix + '.weight').uniform(0.5, 1.5, size=(c,)).astype(np.float32)
    sd[prefix + '.bias'] = _rng_for(seed, prefix + '.bias').uniform(-0.2, 0.2, size=(c,)).astype(np.float32)
    sd[prefix + '.running_mean'] = _rng_for(seed, prefix + '.rm').uniform(-0.3, 0.3, size=(c,)).astype(np.float32)
    sd[prefix + '.running_var'] = _rng_for(seed, prefix + '.rv').uniform(0.5, 1.5, size=(c,)).astype(np.float32)
    sd[prefix + '.num_batches_tracked'] = np.array(0, dtype=np.int64)


MATCHING_BIN_SCORE = 30.0      # dustbin score that goes with make_state_dict(style='matching'): unrelated keypoints fall into the dustbin


def _trained_like(sd, seed, prefix, cout, cin, gain=1.0, rank=24, outliers=6, zero_bias=False):
    """a conv whose statistics look like a TRAINED layer's rather than an i.i.d. draw: a low-rank component that carries most of
    the energy, a dense small remainder, a few outlier output channels with 3x larger rows, and non-zero biases everywhere
    (``zero_bias`` keeps the structural zeros of nets/layers.py:86,145)"""
    g = _rng_for(seed, prefix + '.trained')
    U = g.standard_normal((cout, rank)) / np.sqrt(rank)
    V = g.standard_normal((rank, cin)) / np.sqrt(cin)
    w = gain * (0.9 * U @ V + 0.35 * g.standard_normal((cout, cin)) / np.sqrt(cin))
    hot = g.permutation(cout)[:outliers]
    w[hot] *= 3.0
    b = 0.3 * gain * g.standard_normal(cout)
    if zero_bias:
        b[:] = 0.0
    sd[prefix + '.weight'] = w.reshape(cout, cin, 1).astype(np.float32)
    sd[prefix + '.bias'] = b.astype(np.float32)


def make_state_dict(config: dict, model: str = 'GM', seed: int = 0, bin_score: float = 1.0,
                    gain: float = 1.0, bias_offset: float = 0.0, style: str = 'uniform', qk_gain=None) -> "OrderedDict[str, np.ndarray]":
    """numpy state_dict with the reference key schema.

    ``style='uniform'`` (default; every fixture of rounds 1-2): uniform(+-gain/sqrt(fan_in)) like torch's default
    Conv1d init, last kenc / MLP biases zero as in nets/layers.py:86,145,191,198.  ``bias_offset`` shifts the biases of
    every conv that feeds an InstanceNorm (keypoint encoder, mlp.0) by +-offset: |channel mean| >> channel std.

    ``style='trained'`` (round 3, parity stress): structured weights - low rank + outlier channels + non-zero biases
    (``_trained_like``), the q / k projections 1.5x larger and the residual branch (mlp.3) scaled by 0.3.  Measured on the oracle
    (GM, L = 9, N = 512): the mean of the largest attention probability per query is 0.03 (first layer) ... 0.24 (last) against
    0.002 ... 0.02 with the uniform weights, whose softmax rows are all but flat.  The gains sit where the arithmetic is still
    well conditioned - two fp32 CPU evaluations (reference vs oracle) agree to 1e-5 in the match scores, fp32 vs fp64 to 2e-5 on
    the matched keypoints: with q / k 2x those become 6e-5 / 5e-5 (half of the 1e-4 bar before an implementation has done anything),
    with 3x and 6x outlier channels the network is chaotic (hundreds of index flips between two fp32 evaluations) and a fixture
    would test luck.  (In every setting one or two UNMATCHED keypoints with scores ~0.01 < p flip their mutual-nearest-neighbour
    status between fp32 and fp64: exact-tie noise below the threshold, the same effect the soak test documents.)

    ``style='matching'`` (round 3, evaluation realism - NOT a trained model): a matcher that WORKS on synthetic pairs whose true
    correspondences have similar descriptors: the keypoint encoder and every GNN layer only perturb the descriptors (last
    convolutions scaled by 0.05) and every final projection is 20 I + noise, so the score matrix is a sharpened descriptor
    similarity; with ``bin_score`` = ``MATCHING_BIN_SCORE`` (30) unrelated keypoints fall into the dustbin (measured on the oracle,
    DGNNS, 600 / 560 keypoints, 237 true correspondences: all 237 found + 11 wrong matches; + 62 wrong at bin_score 26, + 213 at 16).  Used by the evaluation tools so that
    precision / pose AUC of the loop (eval/eval_imp.py:213-227) measure the pipeline rather than noise."""
    if style not in ('uniform', 'trained', 'matching'):
        raise ValueError(f'unknown weight style {style!r}')
    cfg = {**DEFAULT_CONFIG, **config}
    D = cfg['descriptor_dim']
    names = cfg['GNN_layers']
    norm = cfg['norm_fn']
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd['bin_score'] = np.array(bin_score, dtype=np.float32)
    chans = [3] + list(cfg['keypoint_encoder']) + [D]
    small = 0.05 if style == 'matching' else 1.0

    def conv(prefix, cout, cin, zero_bias=False, g=1.0, off=0.0):
        if style == 'trained' and cin >= 32:
            _trained_like(sd, seed, prefix, cout, cin, gain=gain * g, zero_bias=zero_bias)
        else:
            _conv(sd, seed, prefix, cout, cin, zero_bias=zero_bias, gain=gain * g, bias_offset=off)

    for i in range(1, len(chans)):
        last = i == len(chans) - 1
        conv(f'kenc.encoder.{3 * (i - 1)}', chans[i], chans[i - 1], zero_bias=last, g=small if last else 1.0,
             off=0.0 if last else bias_offset)
        if not last and norm == 'bn':
            _bn(sd, seed, f'kenc.encoder.{3 * (i - 1) + 1}', chans[i])
    shared = sharing_pattern(len(names), model)
    # gain of the q / k projections (None: 1.5 for 'trained', 1 otherwise): THE conditioning knob - sharper attention rows amplify every
    # rounding of the logits (tools/parity_vs_conditioning.py sweeps it from 1 to 3)
    qk = (1.5 if style == 'trained' else 1.0) if qk_gain is None else float(qk_gain)
    for li in range(len(names)):
        p = f'gnn.layers.{li}'
        if shared[li]:
            conv(p + '.proj', D, D)
            conv(p + '.merge', D, D)
        else:
            conv(p + '.attn.merge', D, D)
            for j in range(3):
                conv(p + f'.attn.proj.{j}', D, D, g=qk if j < 2 else 1.0)
        conv(p + '.mlp.0', 2 * D, 2 * D, off=bias_offset)
        if norm == 'bn':
            _bn(sd, seed, p + '.mlp.1', 2 * D)
        conv(p + '.mlp.3', D, 2 * D, zero_bias=True, g=small * (0.3 if style == 'trained' else 1.0))
    for i in range(cfg['n_layers']):
        conv(f'final_proj.{i}', D, D)
        if style == 'matching':
            w = sd[f'final_proj.{i}.weight']
            sd[f'final_proj.{i}.weight'] = (0.5 * w + 20.0 * np.eye(D, dtype=np.float32)[:, :, None]).astype(np.float32)
            sd[f'final_proj.{i}.bias'] = (0.0 * sd[f'final_proj.{i}.bias']).astype(np.float32)
    return sd


def make_pair(n0: int, n1: int, desc_dim: int = 256, seed: int = 0, batch: int = 1,
              width: int = 640, height: int = 480):
    """SuperPoint-like synthetic inputs: kpts ~ U(image), scores ~ U(0,1),
    descriptors = L2-normalised N(0,1).  Returns a dict of float32 numpy arrays with the
    reference's ``data`` keys (nets/gm.py:147-149) plus ``image_shape`` = (B,3,H,W)."""
    out = {}
    for side, n in ((0, n0), (1, n1)):
        g = _rng_for(seed, f'pair.side{side}')
        kp = g.uniform(0.0, 1.0, size=(batch, n, 2)) * np.array([width, height], dtype=np.float64)
        sc = g.uniform(0.0, 1.0, size=(batch, n))
        de = g.standard_normal(size=(batch, n, desc_dim))
        de /= np.maximum(np.linalg.norm(de, axis=-1, keepdims=True), 1e-12)
        out[f'keypoints{side}'] = kp.astype(np.float32)
        out[f'scores{side}'] = sc.astype(np.float32)
        out[f'descriptors{side}'] = de.astype(np.float32)
    out['image_shape'] = (batch, 3, height, width)
    return out


def make_correlated_pair(n0: int, n1: int, desc_dim: int = 256, seed: int = 0, batch: int = 1,
                         overlap: float = 0.6, noise: float = 0.25, width: int = 640, height: int = 480):
    """Like make_pair but image 1 re-observes a fraction of image 0's keypoints (shifted position,
    perturbed descriptor) so that a matcher with random weights still sees structure: used to get
    non-trivial match sets / pooling behaviour in tests."""
    out = make_pair(n0, n1, desc_dim, seed, batch, width, height)
    g = _rng_for(seed, 'pair.corr')
    k = int(min(n0, n1) * overlap)
    for b in range(batch):
        src = g.permutation(n0)[:k]
        dst = g.permutation(n1)[:k]
        d = out['descriptors0'][b, src] + noise * g.standard_normal(size=(k, desc_dim)).astype(np.float32) / np.sqrt(desc_dim)
        d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-12)
        out['descriptors1'][b, dst] = d.astype(np.float32)
        sh = out['keypoints0'][b, src] + g.normal(0, 2.0, size=(k, 2)).astype(np.float32) + np.array([12.0, -7.0], dtype=np.float32)
        sh[:, 0] = np.clip(sh[:, 0], 0, width - 1)
        sh[:, 1] = np.clip(sh[:, 1], 0, height - 1)
        out['keypoints1'][b, dst] = sh
        out['scores1'][b, dst] = np.clip(out['scores0'][b, src] + 0.05 * g.standard_normal(size=k), 0.01, 0.99).astype(np.float32)
    return out


def make_two_view_pair(n0: int, n1: int, desc_dim: int = 256, seed: int = 0, overlap: float = 0.6, noise_px: float = 0.5,
                       desc_noise: float = 0.25, angle_deg: float = 12.0, width: int = 640, height: int = 480, outlier_ratio: float = 0.0):
    """A pair with GEOMETRY: image 0 sees random 3D points, image 1 re-observes a fraction ``overlap`` of them from a second camera
    (rotation ``angle_deg`` about a random axis, unit baseline, pixel noise ``noise_px``, descriptors perturbed like
    :func:`make_correlated_pair`); every other keypoint of either image is an unrelated distractor.  On such pairs the pose step and
    the metrics tail of the evaluation loop (eval/eval_imp.py:112-141) measure something: returns the ``data`` keys of
    :func:`make_pair` (batch 1) plus ``K0, K1`` (3x3), ``T_0to1`` (3x4 [R|t], |t| = 1) and the ground-truth essential matrix ``E``
    in intrinsics-normalised coordinates, as the reference's dumps carry them (components/readers.py:14-33)."""
    out = make_pair(n0, n1, desc_dim, seed, 1, width, height)
    g = _rng_for(seed, 'pair.twoview')
    K = np.array([[520., 0, width / 2.], [0, 520., height / 2.], [0, 0, 1.]])
    ax = g.normal(size=3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(angle_deg)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
    t = g.normal(size=3); t /= np.linalg.norm(t)
    k = int(min(n0, n1) * overlap)
    src = g.permutation(n0)[:k]
    dst = g.permutation(n1)[:k]
    # 3D points behind the chosen keypoints of image 0 (depth 4..9), seen again in image 1 when they project inside it
    x0 = out['keypoints0'][0, src].astype(np.float64)
    depth = g.uniform(4.0, 9.0, size=k)
    X = np.concatenate([(x0 - K[:2, 2]) / np.diag(K)[:2], np.ones((k, 1))], 1) * depth[:, None]
    Xc = X @ R.T + t
    x1 = Xc[:, :2] / Xc[:, 2:] * np.diag(K)[:2] + K[:2, 2]
    x1 += g.normal(0, noise_px, size=x1.shape)
    ok = (Xc[:, 2] > 0.1) & (x1[:, 0] >= 0) & (x1[:, 0] < width) & (x1[:, 1] >= 0) & (x1[:, 1] < height)
    src, dst, x1 = src[ok], dst[ok], x1[ok]
    d = out['descriptors0'][0, src] + desc_noise * g.standard_normal(size=(len(src), desc_dim)).astype(np.float32) / np.sqrt(desc_dim)
    d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-12)
    out['descriptors1'][0, dst] = d.astype(np.float32)
    out['keypoints1'][0, dst] = x1.astype(np.float32)
    out['scores1'][0, dst] = np.clip(out['scores0'][0, src] + 0.05 * g.standard_normal(size=len(src)), 0.01, 0.99).astype(np.float32)
    if outlier_ratio > 0.0:
        # look-alikes (round 4: the evaluation set must not be a best case): further keypoints of image 0 get a descriptor twin at an
        # UNRELATED position of image 1 - a descriptor-driven matcher pairs them up, the geometry says no.  `outlier_ratio` = their share
        # of all planted correspondences
        used0, used1 = np.zeros(n0, bool), np.zeros(n1, bool)
        used0[src] = True; used1[dst] = True
        free0, free1 = np.nonzero(~used0)[0], np.nonzero(~used1)[0]
        m = int(round(len(src) * outlier_ratio / max(1e-9, 1.0 - outlier_ratio)))
        m = min(m, len(free0), len(free1))
        a = g.permutation(free0)[:m]
        b = g.permutation(free1)[:m]
        dd = out['descriptors0'][0, a] + desc_noise * g.standard_normal(size=(m, desc_dim)).astype(np.float32) / np.sqrt(desc_dim)
        dd /= np.maximum(np.linalg.norm(dd, axis=-1, keepdims=True), 1e-12)
        out['descriptors1'][0, b] = dd.astype(np.float32)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    out.update({'K0': K.copy(), 'K1': K.copy(), 'T_0to1': np.hstack([R, t.reshape(3, 1)]), 'E': tx @ R,
                'true_matches': np.stack([src, dst], 1)})
    return out


def make_hard_two_view_pair(seed: int, n_lo: int = 1000, n_hi: int = 2048, **kw):
    """one pair of the HARDER evaluation set of round 4 (the 64 scenes of round 3 all left the loop at the earliest
    possible iteration): keypoint counts ~ U(n_lo, n_hi) per image, overlap ~ U(0.2, 0.8), pixel noise ~ U(0.5, 2), look-alike
    outliers ~ U(0.3, 0.7) of the planted correspondences, rotation ~ U(5, 25) degrees - all drawn from the pair's seed"""
    g = _rng_for(seed, 'pair.hard')
    n0, n1 = int(g.integers(n_lo, n_hi + 1)), int(g.integers(n_lo, n_hi + 1))
    return make_two_view_pair(n0, n1, seed=seed, overlap=float(g.uniform(0.2, 0.8)), noise_px=float(g.uniform(0.5, 2.0)),
                              outlier_ratio=float(g.uniform(0.3, 0.7)), angle_deg=float(g.uniform(5.0, 25.0)), **kw)


class PoseStub:
    """Deterministic stand-in for ``eval/pose_estimation.py:92-115 estimate_pose`` (cv2 MAGSAC is absent here) with the
    reference's keyword signature, so that the SAME object can drive the imported reference loop, the oracle loop and
    the HIP loop through ``eval/matching.py:84-117`` (pose-change test, early exit with inlier-filtered indices) and
    ``:243-252`` (``with_uncertainty``: ``mscore_th = 0.2 * inlier_ratio``).

    Call k returns ``None`` when ``schedule[k] is None``; otherwise ``(E, R, t, inliers)`` with R = rotation by
    ``schedule[k]`` degrees about z, t = R @ [1, 0, 0] and an inlier mask that is a pure function of the matched
    coordinates (so it is identical for any implementation that passes the same matches).  After the schedule is
    exhausted the last entry repeats."""

    def __init__(self, schedule, keep_mod=10, keep_below=8):
        self.schedule, self.keep_mod, self.keep_below = list(schedule), keep_mod, keep_below
        self.calls = []

    def __call__(self, kpts0, kpts1, K0=None, K1=None, norm_thresh=1.0, method=None, **kw):
        k = len(self.calls)
        ang = self.schedule[min(k, len(self.schedule) - 1)]
        kpts0, kpts1 = np.asarray(kpts0), np.asarray(kpts1)
        self.calls.append((kpts0.shape[0], ang))
        if ang is None:
            return None
        a = np.deg2rad(float(ang))
        R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        t = R @ np.array([1.0, 0.0, 0.0])
        h = (np.floor(kpts0[:, 0]).astype(np.int64) * 31 + np.floor(kpts1[:, 1]).astype(np.int64) * 17
             + np.floor(kpts0[:, 1]).astype(np.int64) * 7) % self.keep_mod
        return np.eye(3), R, t, h < self.keep_below


SUPERPOINT_LAYERS = [('conv1a', 1, 64, 3), ('conv1b', 64, 64, 3), ('conv2a', 64, 64, 3), ('conv2b', 64, 64, 3),
                     ('conv3a', 64, 128, 3), ('conv3b', 128, 128, 3), ('conv4a', 128, 128, 3), ('conv4b', 128, 128, 3),
                     ('convPa', 128, 256, 3), ('convPb', 256, 65, 1), ('convDa', 128, 256, 3), ('convDb', 256, 256, 1)]


def make_superpoint_state_dict(seed: int = 0, descriptor_dim: int = 256, gain: float = 2.449, head_gain: float = 4.0):
    """numpy state_dict with the key schema of nets/superpoint.py:120-137 (conv1a ... convDb, weight [out, in, k, k] + bias);
    uniform(+-gain / sqrt(fan_in)): gain sqrt(6) keeps the activation scale through the ReLU stack; the last detector conv is
    scaled by head_gain so the 65-way softmax is peaked and the keypoint threshold actually cuts (superpoint_v1.pth is not
    available here)."""
    sd = OrderedDict()
    for name, cin, cout, k in SUPERPOINT_LAYERS:
        if name == 'convDb':
            cout = descriptor_dim
        bound = gain * (head_gain if name == 'convPb' else 1.0) / np.sqrt(cin * k * k)
        sd[name + '.weight'] = _rng_for(seed, 'sp.' + name + '.weight').uniform(-bound, bound, size=(cout, cin, k, k)).astype(np.float32)
        sd[name + '.bias'] = _rng_for(seed, 'sp.' + name + '.bias').uniform(-bound, bound, size=(cout,)).astype(np.float32)
    return sd


def make_image(height: int, width: int, seed: int = 0, batch: int = 1):
    """grayscale test image in (0, 1), float32 [B, 1, H, W]: random blobs and edges + fine noise, squashed with tanh instead of
    clipped so that no region is exactly flat (flat regions give exactly equal detector scores, whose top-k order is arbitrary)"""
    g = _rng_for(seed, 'image')
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    out = np.zeros((batch, 1, height, width))
    for b in range(batch):
        img = 0.08 * g.standard_normal((height, width))
        for _ in range(40):
            cx, cy, r, a = g.uniform(0, width), g.uniform(0, height), g.uniform(3, 25), g.uniform(-0.5, 0.5)
            img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
        for _ in range(10):
            th, off, a = g.uniform(0, np.pi), g.uniform(-100, 100), g.uniform(-0.3, 0.3)
            img += a * ((xx * np.cos(th) + yy * np.sin(th) + off) > (width + height) / 4)
        out[b, 0] = 0.5 + 0.45 * np.tanh(1.5 * img)
    return out.astype(np.float32)
