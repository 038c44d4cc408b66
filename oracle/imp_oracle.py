"""CPU ORACLE for the IMP / EIMP matching hot path.   *** TEST INFRASTRUCTURE ONLY ***

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path (``imp-release_amd``) never
falls back to it and fails loudly when the HIP library is missing.

It is an independent restatement (torch-CPU fp32, token-major ``[B, N, C]`` tensors, plain
functions over a ``state_dict``) of the reference algorithm; every function cites the reference
``file:line`` it follows (paths relative to the reference repo root).

PARITY PIN: validated in the build container against the *imported* reference
(``tools/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks
the oracle against those vectors on every run).  The reference holds no tests or golden vectors
of its own, so the reference-run outputs are the pin.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

EPS = 1e-8          # nets/layers.py:13
IN_EPS = 1e-3       # nets/layers.py:68,70
NUM_HEADS = 4       # hard-coded in nets/layers.py:157,230
VALID_ITS = (3, 5, 7, 9, 11, 13, 14)   # eval/matching.py:43,154


# --------------------------------------------------------------------------- helpers
def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x)


def to_torch_state_dict(sd) -> Dict[str, torch.Tensor]:
    return {k: _t(v).clone() for k, v in sd.items()}


def sharing_pattern(n_gnn_layers: int, model: str) -> List[bool]:
    """nets/gms.py:17 and nets/adgm.py:18: [F,F]*2 + [F,F,T,T]*21 ; GM (nets/gm.py:62) shares nothing."""
    if model == 'GM':
        return [False] * n_gnn_layers
    return ([False, False] * 2 + [False, False, True, True] * 21)[:n_gnn_layers]


# --------------------------------------------------------------------------- ops (nets/layers.py)
def normalize_keypoints(kpts: torch.Tensor, image_shape: Sequence[int]) -> torch.Tensor:
    """nets/layers.py:49-56: centre on [w,h]/2 and divide by 0.7*max(w,h); (h,w) = image_shape[2:4]."""
    _, _, height, width = image_shape
    size = torch.tensor([float(width), float(height)], dtype=kpts.dtype)
    center = size / 2
    scaling = size.max() * 0.7
    return (kpts - center) / scaling


def conv1x1(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Conv1d(kernel_size=1) on token-major data: y[b,n,:] = W x[b,n,:] + bias."""
    w = sd[prefix + '.weight'][:, :, 0]
    return x @ w.t() + sd[prefix + '.bias']


def _norm(sd, prefix: str, x: torch.Tensor, norm_fn: str) -> torch.Tensor:
    if norm_fn == 'in':
        # nn.InstanceNorm1d(C, eps=1e-3): per (b, channel) mean / biased variance over the N tokens,
        # no affine, instance statistics also in eval mode (nets/layers.py:67-68)
        mean = x.mean(dim=1, keepdim=True)
        var = x.var(dim=1, unbiased=False, keepdim=True)
        return (x - mean) / torch.sqrt(var + IN_EPS)
    if norm_fn == 'bn':
        # nn.BatchNorm1d(C, eps=1e-3) in eval mode (nets/layers.py:69-70)
        rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
        return (x - rm) / torch.sqrt(rv + IN_EPS) * sd[prefix + '.weight'] + sd[prefix + '.bias']
    raise ValueError(norm_fn)


def _act(x: torch.Tensor, ac_fn: str) -> torch.Tensor:
    if ac_fn == 'relu':
        return torch.relu(x)
    if ac_fn == 'gelu':
        return torch.nn.functional.gelu(x)
    if ac_fn == 'lrelu':
        return torch.nn.functional.leaky_relu(x, 0.1)
    raise ValueError(ac_fn)


def mlp(sd, prefix: str, n_convs: int, x: torch.Tensor, norm_fn: str, ac_fn: str) -> torch.Tensor:
    """nets/layers.py:59-77: Sequential(conv, norm, act, conv, norm, act, ..., conv); conv j sits at index 3j."""
    for j in range(n_convs):
        x = conv1x1(sd, f'{prefix}.{3 * j}', x)
        if j < n_convs - 1:
            x = _act(_norm(sd, f'{prefix}.{3 * j + 1}', x, norm_fn), ac_fn)
    return x


def keypoint_encoder(sd, cfg, nkpts: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
    """nets/layers.py:80-90: cat(x, y, score) -> MLP [3,32,64,128,256,D]."""
    inp = torch.cat([nkpts, scores.unsqueeze(-1)], dim=-1)
    return mlp(sd, 'kenc.encoder', len(cfg['keypoint_encoder']) + 1, inp, cfg['norm_fn'], cfg['ac_fn'])


def _split_heads(x: torch.Tensor) -> torch.Tensor:
    """nets/layers.py:119-120: ``.view(B, dim, heads, N)`` on channel-major data means channel
    c = d*heads + h.  Token-major [B,N,C] -> [B, H, N, d]."""
    B, N, C = x.shape
    return x.view(B, N, C // NUM_HEADS, NUM_HEADS).permute(0, 3, 1, 2)


def _merge_heads(x: torch.Tensor) -> torch.Tensor:
    """inverse of _split_heads: [B,H,N,d] -> [B,N,C] with c = d*H + h (nets/layers.py:134)."""
    B, H, N, d = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, N, d * H)


def attention_prob(q: torch.Tensor, k: torch.Tensor, key_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """nets/layers.py:121-129: softmax_m(q.k / sqrt(d)); masked keys filled with -FLT_MAX first.
    key_mask: [B, N, M] of {0,1} as in the reference (M[:, None] broadcast over heads)."""
    d = q.shape[-1]
    s = torch.einsum('bhnd,bhmd->bhnm', q, k) / d ** .5
    if key_mask is not None:
        s = s.masked_fill((1 - key_mask[:, None]).bool(), -torch.finfo(s.dtype).max)
    return torch.softmax(s, dim=-1)


def propagate(sd, cfg, li: int, shared: bool, x: torch.Tensor, src: torch.Tensor,
              prob: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None
              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Shared)AttentionalPropagation: nets/layers.py:139-149 and :182-218.  Returns (delta, prob)."""
    p = f'gnn.layers.{li}'
    if not shared:
        q = _split_heads(conv1x1(sd, p + '.attn.proj.0', x))
        k = _split_heads(conv1x1(sd, p + '.attn.proj.1', src))
        v = _split_heads(conv1x1(sd, p + '.attn.proj.2', src))
        prob = attention_prob(q, k, mask)
        msg = conv1x1(sd, p + '.attn.merge', _merge_heads(prob @ v))
    else:
        v = _split_heads(conv1x1(sd, p + '.proj', src))
        msg = conv1x1(sd, p + '.merge', _merge_heads(prob @ v))
    delta = mlp(sd, p + '.mlp', 2, torch.cat([x, msg], dim=-1), cfg['norm_fn'], cfg['ac_fn'])
    return delta, prob


def dustbin_augment(M: torch.Tensor, dustbin: torch.Tensor) -> torch.Tensor:
    """nets/layers.py:39-40 (same in dual_softmax :21-22): append a dustbin column then a dustbin row."""
    B, n, m = M.shape
    M = torch.cat([M, dustbin.expand(B, n, 1)], dim=-1)
    return torch.cat([M, dustbin.expand(B, 1, m + 1)], dim=-2)


def sinkhorn(M_aug: torch.Tensor, iteration: int) -> torch.Tensor:
    """nets/layers.py:27-46: probability-domain Sinkhorn on a row softmax; marginals
    r = [1,...,1, N+1], c = [1,...,1, M+1] (own dim + 1, nets/layers.py:41-44); returns p*u*v."""
    B, n1, m1 = M_aug.shape
    r = torch.ones(B, n1)
    r[:, -1] = n1
    c = torch.ones(B, m1)
    c[:, -1] = m1
    p = torch.softmax(M_aug, dim=-1)
    u = torch.ones_like(r)
    v = torch.ones_like(c)
    for _ in range(iteration):
        u = r / ((p * v.unsqueeze(-2)).sum(-1) + EPS)
        v = c / ((p * u.unsqueeze(-1)).sum(-2) + EPS)
    return p * u.unsqueeze(-1) * v.unsqueeze(-2)


def dual_softmax(M_aug: torch.Tensor) -> torch.Tensor:
    """nets/layers.py:20-24."""
    return torch.exp(torch.log_softmax(M_aug, dim=-1) + torch.log_softmax(M_aug, dim=1))


def compute_matches(scores: torch.Tensor, p: float):
    """nets/gm.py:305-320: mutual nearest neighbours on the inner N x M block, threshold p, -1 = unmatched."""
    inner = scores[:, :-1, :-1]
    max0, max1 = inner.max(2), inner.max(1)
    i0, i1 = max0.indices, max1.indices
    ar0 = torch.arange(i0.shape[1])[None]
    ar1 = torch.arange(i1.shape[1])[None]
    mutual0 = ar0 == i1.gather(1, i0)
    mutual1 = ar1 == i0.gather(1, i1)
    zero = scores.new_tensor(0)
    ms0 = torch.where(mutual0, max0.values, zero)
    ms1 = torch.where(mutual1, ms0.gather(1, i1), zero)
    valid0 = mutual0 & (ms0 > p)
    valid1 = mutual1 & valid0.gather(1, i1)
    return (torch.where(valid0, i0, i0.new_tensor(-1)), torch.where(valid1, i1, i1.new_tensor(-1)), ms0, ms1)


def attention_received(prob: torch.Tensor) -> torch.Tensor:
    """nets/adgm.py:557-565: attention mass received per key, summed over heads and queries, L1-normalised."""
    s = prob.sum(dim=1).sum(dim=1)
    return s / s.sum(dim=1, keepdim=True)


def _pool_side(mass: torch.Tensor, a_self: torch.Tensor, a_cross: torch.Tensor, thr: float):
    """one side of nets/adgm.py:577-589: confident rows, lower-median thresholds, sorted union."""
    pids = torch.where(mass >= thr)[0]
    if pids.numel() == 0:
        return None
    md_s = torch.median(a_self[pids])      # torch.median = LOWER median for even counts
    md_c = torch.median(a_cross[pids])
    return torch.unique(torch.hstack([pids, torch.where(a_self >= md_s)[0], torch.where(a_cross >= md_c)[0]]))


def pool(pred_score, prob00, prob01, prob11, prob10, mscore_th=0.1, uncertainty_ratio=1.0, n_min_tokens=256):
    """AdaGMN.pool, nets/adgm.py:552-605 (batch element 0 only, as the reference).
    prob00 [1,4,N,N]; prob01 [1,4,M,N] (image-1 queries over image-0 keys); prob11 [1,4,M,M]; prob10 [1,4,N,M].
    NOTE n0/n1 are the *augmented* sizes (N+1, M+1): nets/adgm.py:554-555,567,571."""
    n0, n1 = pred_score.shape[1], pred_score.shape[2]
    a00, a01 = attention_received(prob00)[0], attention_received(prob01)[0]
    a10, a11 = attention_received(prob10)[0], attention_received(prob11)[0]
    thr = mscore_th * uncertainty_ratio
    inner = pred_score[0, :-1, :-1]
    ids0 = ids1 = None
    if not (n_min_tokens > 0 and n0 <= n_min_tokens):
        ids0 = _pool_side(inner.sum(dim=-1), a00, a01, thr)
    if not (n_min_tokens > 0 and n1 <= n_min_tokens):
        ids1 = _pool_side(inner.sum(dim=0), a11, a10, thr)
    return ids0, ids1


# --------------------------------------------------------------------------- model-level oracle
class MatcherOracle:
    """GM / DGNNS / AdaGMN inference surface over a reference-schema state_dict."""

    def __init__(self, config: dict, state_dict, model: str = 'GM'):
        from_defaults = {
            'descriptor_dim': 256, 'keypoint_encoder': [32, 64, 128, 256], 'GNN_layers': ['self', 'cross'] * 9,
            'sinkhorn_iterations': 20, 'match_threshold': 0.2, 'n_layers': 9, 'n_min_tokens': 256,
            'with_sinkhorn': True, 'ac_fn': 'relu', 'norm_fn': 'bn'}       # nets/gm.py:30-44
        self.config = {**from_defaults, **config}
        assert model in ('GM', 'DGNNS', 'AdaGMN')
        self.model = model
        self.sd = to_torch_state_dict(state_dict)
        self.names = list(self.config['GNN_layers'])
        self.shared = sharing_pattern(len(self.names), model)
        self.sinkhorn_iterations = self.config['sinkhorn_iterations']
        self.bin_score = self.sd['bin_score']
        self.self_prob0 = self.self_prob1 = self.cross_prob0 = self.cross_prob1 = None

    # -- step API (eval/matching.py:47-61,158,176-193,254) -------------------------------------
    def encode_keypoint(self, norm_kpts0, norm_kpts1, scores0, scores1):
        """nets/gm.py:287-288"""
        return (keypoint_encoder(self.sd, self.config, norm_kpts0, scores0),
                keypoint_encoder(self.sd, self.config, norm_kpts1, scores1))

    def forward_one_layer(self, desc0, desc1, layer_i):
        """nets/gm.py:263-285 (GM: never shares) / nets/gms.py:260-282 / nets/adgm.py:528-550."""
        sh = self.shared[layer_i]
        if self.names[layer_i] == 'cross':
            d0, self.cross_prob1 = propagate(self.sd, self.config, layer_i, sh, desc0, desc1, self.cross_prob1)
            d1, self.cross_prob0 = propagate(self.sd, self.config, layer_i, sh, desc1, desc0, self.cross_prob0)
        else:
            d0, self.self_prob0 = propagate(self.sd, self.config, layer_i, sh, desc0, desc0, self.self_prob0)
            d1, self.self_prob1 = propagate(self.sd, self.config, layer_i, sh, desc1, desc1, self.self_prob1)
        return desc0 + d0, desc1 + d1

    def compute_distance(self, desc0, desc1, layer_id=-1):
        """nets/gm.py:290-295"""
        idx = layer_id if layer_id >= 0 else self.config['n_layers'] + layer_id
        m0 = conv1x1(self.sd, f'final_proj.{idx}', desc0)
        m1 = conv1x1(self.sd, f'final_proj.{idx}', desc1)
        return torch.einsum('bnd,bmd->bnm', m0, m1) / self.config['descriptor_dim'] ** .5

    def compute_score(self, dist, iteration=None):
        """nets/gm.py:297-303"""
        it = self.sinkhorn_iterations if iteration is None else iteration
        aug = dustbin_augment(dist, self.bin_score)
        return sinkhorn(aug, it) if self.config['with_sinkhorn'] else dual_softmax(aug)

    compute_matches = staticmethod(compute_matches)

    def pool(self, pred_score, prob00, prob01, prob11, prob10, mscore_th=0.1, uncertainty_ratio=1.0,
             n_min_tokens=256):
        if self.model != 'AdaGMN':
            return None, None                      # nets/gms.py:316-317
        return pool(pred_score, prob00, prob01, prob11, prob10, mscore_th, uncertainty_ratio, n_min_tokens)

    # -- whole-pair drivers --------------------------------------------------------------------
    def _prepare(self, data):
        """shared head of produce_matches: nets/gm.py:147-178 (normalisation + encoder + residual add)."""
        desc0, desc1 = data['descriptors0'], data['descriptors1']
        if 'norm_keypoints0' in data and 'norm_keypoints1' in data:
            nk0, nk1 = data['norm_keypoints0'], data['norm_keypoints1']
        elif 'image0' in data and 'image1' in data:
            nk0 = normalize_keypoints(data['keypoints0'], data['image0'].shape)
            nk1 = normalize_keypoints(data['keypoints1'], data['image1'].shape)
        else:
            raise ValueError('Require image shape for keypoint coordinate normalization')
        e0, e1 = self.encode_keypoint(nk0, nk1, data['scores0'], data['scores1'])
        return desc0 + e0, desc1 + e1

    def _score_and_match(self, desc0, desc1, it, p):
        dist = self.compute_distance(desc0, desc1, it)
        score = self.compute_score(dist)
        return (score,) + compute_matches(score, p)

    def produce_matches(self, data, p=0.2, only_last=False, mscore_th=0.1, uncertainty_ratio=1.0):
        if self.model == 'AdaGMN':
            return self._produce_matches_ada(data, p, mscore_th, uncertainty_ratio)
        desc0, desc1 = self._prepare(data)
        nI = self.config['n_layers']
        out = {'scores': [], 'indices0': [], 'indices1': [], 'mscores0': [], 'mscores1': [],
               'prob00': [], 'prob01': [], 'prob11': [], 'prob10': []}
        for it in range(nI):
            # nets/layers.py:161-179 (GM) and nets/gms.py:189-217 (DGNNS): self pair then cross pair,
            # both deltas of a pair computed from the pre-update descriptors.
            desc0, desc1 = self.forward_one_layer(desc0, desc1, 2 * it)
            desc0, desc1 = self.forward_one_layer(desc0, desc1, 2 * it + 1)
            if self.model == 'DGNNS':
                out['prob00'].append(self.self_prob0); out['prob11'].append(self.self_prob1)
                out['prob10'].append(self.cross_prob1); out['prob01'].append(self.cross_prob0)
            if (not only_last) or it == nI - 1:
                # nets/gm.py:185-204 (GM stacks iterations on the batch axis; per-iteration here) and
                # nets/gms.py:224-248
                s, i0, i1, m0, m1 = self._score_and_match(desc0, desc1, it, p)
                out['scores'].append(s); out['indices0'].append(i0); out['indices1'].append(i1)
                out['mscores0'].append(m0); out['mscores1'].append(m1)
        return out

    def _produce_matches_ada(self, data, p, mscore_th, uncertainty_ratio):
        """AdaGMN.produce_matches, nets/adgm.py:327-526: *masked* adaptive pooling (tensors never shrink)."""
        desc0, desc1 = self._prepare(data)
        cfg = self.config
        nB, nK0, nK1 = desc0.shape[0], desc0.shape[1], desc1.shape[1]
        nI = cfg['n_layers']
        n_min = cfg['n_min_tokens']
        gids0 = [torch.arange(nK0) for _ in range(nB)]
        gids1 = [torch.arange(nK1) for _ in range(nB)]
        M00 = M01 = M10 = M11 = None
        p00 = p01 = p11 = p10 = None
        all_i0, all_m0, pred_score = [], [], None
        first_it_to_update = 2                                                    # nets/adgm.py:32
        for ni in range(nI):
            sh_s, sh_c = self.shared[2 * ni], self.shared[2 * ni + 1]
            d0, p00 = propagate(self.sd, cfg, 2 * ni, sh_s, desc0, desc0, p00, M00)
            d1, p11 = propagate(self.sd, cfg, 2 * ni, sh_s, desc1, desc1, p11, M11)
            desc0, desc1 = desc0 + d0, desc1 + d1
            mc10 = None if (M10 is None or ni == 3) else M10                     # nets/adgm.py:392,396
            mc01 = None if (M01 is None or ni == 3) else M01
            d0, p10 = propagate(self.sd, cfg, 2 * ni + 1, sh_c, desc0, desc1, p10, mc10)
            d1, p01 = propagate(self.sd, cfg, 2 * ni + 1, sh_c, desc1, desc0, p01, mc01)
            desc0, desc1 = desc0 + d0, desc1 + d1
            md0 = conv1x1(self.sd, f'final_proj.{ni}', desc0)
            md1 = conv1x1(self.sd, f'final_proj.{ni}', desc1)
            if ni < first_it_to_update:
                dist = torch.einsum('bnd,bmd->bnm', md0, md1) / cfg['descriptor_dim'] ** .5
                pred_score = self.compute_score(dist)
                i0, _, m0, _ = compute_matches(pred_score, p)
                all_i0.append(i0); all_m0.append(m0)
                continue
            b_i0 = torch.full((nB, nK0), -1, dtype=torch.long)
            b_m0 = torch.zeros(nB, nK0)
            updating = self.shared[2 * ni]                                        # nets/adgm.py:422
            if updating:
                a00, a01 = attention_received(p00), attention_received(p01)
                a10, a11 = attention_received(p10), attention_received(p11)
                M00 = torch.zeros(nB, nK0, nK0); M01 = torch.zeros(nB, nK1, nK0)
                M11 = torch.zeros(nB, nK1, nK1); M10 = torch.zeros(nB, nK0, nK1)
            for bi in range(nB):
                g0, g1 = gids0[bi], gids1[bi]
                dist = torch.einsum('bnd,bmd->bnm', md0[bi, g0][None], md1[bi, g1][None]) / cfg['descriptor_dim'] ** .5
                pred_score = self.compute_score(dist)
                i0, _, m0, _ = compute_matches(pred_score, p)
                i0, m0 = i0[0], m0[0]
                v0 = i0 >= 0
                b_i0[bi, g0[v0]] = g1[i0[v0]]
                b_m0[bi, g0] = m0
                if updating:
                    thr = mscore_th * uncertainty_ratio
                    inner = pred_score[0, :-1, :-1]
                    if not (n_min > 0 and g0.shape[-1] <= n_min):                 # nets/adgm.py:465 (N, not N+1)
                        f0 = _pool_side(inner.sum(-1), a00[bi][g0], a01[bi][g0], thr)
                        if f0 is not None:
                            g0 = g0[f0]
                    if not (n_min > 0 and g1.shape[-1] <= n_min):
                        f1 = _pool_side(inner.sum(0), a11[bi][g1], a10[bi][g1], thr)
                        # NOTE nets/adgm.py:491-494 uses norm_prob10 for "md_prob10"/aug_ids10 first and
                        # norm_prob11 second; the union is order-independent.
                        if f1 is not None:
                            g1 = g1[f1]
                    gids0[bi], gids1[bi] = g0, g1
                    M00[bi][:, g0] = 1; M01[bi][:, g0] = 1
                    M11[bi][:, g1] = 1; M10[bi][:, g1] = 1
            all_i0.append(b_i0); all_m0.append(b_m0)
        return {'scores': [pred_score], 'indices0': all_i0, 'mscores0': all_m0}

    def run(self, data):
        """nets/gm.py:322-364 (GM -> {'p'}) ; nets/gms.py:284-314, nets/adgm.py:607-635 (-> index0/index1)."""
        d = {'descriptors0': data['desc1'], 'descriptors1': data['desc2'],
             'norm_keypoints0': data['x1'][:, :, :2], 'norm_keypoints1': data['x2'][:, :, :2],
             'scores0': data['x1'][:, :, -1], 'scores1': data['x2'][:, :, -1]}
        if self.model == 'GM':
            desc0, desc1 = self._prepare(d)
            for li in range(len(self.names)):
                desc0, desc1 = self.forward_one_layer(desc0, desc1, li)
            return {'p': self.compute_score(self.compute_distance(desc0, desc1, -1))}
        out = self.produce_matches(d, p=self.config['match_threshold'], only_last=True)
        i0 = out['indices0'][-1][0]
        index0 = torch.where(i0 >= 0)[0]
        return {'index0': index0, 'index1': i0[index0]}


# --------------------------------------------------------------------------- iterative loops
def _angle_mat(R1, R2):
    """tools/utils.py:425-428 (rotation angle between two rotations, degrees)"""
    import numpy as np
    c = (np.trace(R1.T @ R2) - 1) / 2
    return float(np.rad2deg(np.abs(np.arccos(np.clip(c, -1., 1.)))))


def _angle_vec(v1, v2):
    """tools/utils.py:431-434"""
    import numpy as np
    n = np.linalg.norm(v1) * np.linalg.norm(v2)
    return float(np.rad2deg(np.arccos(np.clip(np.dot(v1, v2) / n, -1.0, 1.0))))


def matching_iterative(data, model: MatcherOracle, nI=15, match_ratio=0.1, min_kpts=25,
                       estimate_pose: Optional[Callable] = None, stop_pose: Optional[float] = 1.5,
                       uncertainty: bool = False, with_uncertainty: bool = False, trace: Optional[list] = None,
                       error_th=1.0, method=None):
    """Host control flow of eval/matching.py:16-123 (uncertainty=False, IMP) and :126-276 (True, EIMP:
    real ragged slicing + pool).  ``estimate_pose`` has the reference's keyword signature
    (eval/pose_estimation.py:92: kpts0, kpts1, K0, K1, norm_thresh, method -> None | (E, R, t, inliers)); the cv2
    MAGSAC solver itself is out of scope - the fixtures drive this slot with
    ``imp_release_amd.synthetic.PoseStub`` so that the pose-change early exit (:84-117) and the
    ``with_uncertainty`` threshold (:243-252) are exercised.  ``estimate_pose=None`` = no pose ever found.
    Returns dict(indices0, mscores0, n_iter, keep0, keep1, R, t) with keep* = surviving original keypoint ids."""
    nk0 = data.get('norm_keypoints0'); nk1 = data.get('norm_keypoints1')
    if nk0 is None:
        nk0 = normalize_keypoints(data['keypoints0'], data['image0'].shape)
        nk1 = normalize_keypoints(data['keypoints1'], data['image1'].shape)
    pts0 = data['keypoints0'][0].numpy(); pts1 = data['keypoints1'][0].numpy()
    e0, e1 = model.encode_keypoint(nk0, nk1, data['scores0'], data['scores1'])
    desc0, desc1 = data['descriptors0'] + e0, data['descriptors1'] + e1
    keep0, keep1 = torch.arange(desc0.shape[1]), torch.arange(desc1.shape[1])
    sel0 = sel1 = None
    last_R = last_t = None
    pred_score = None
    for it in range(nI):
        if uncertainty:
            if sel0 is not None:                                  # eval/matching.py:166-169
                desc0, keep0 = desc0[:, sel0], keep0[sel0]
            if sel1 is not None:                                  # eval/matching.py:171-174
                desc1, keep1 = desc1[:, sel1], keep1[sel1]
            sel0 = sel1 = None
        desc0, desc1 = model.forward_one_layer(desc0, desc1, 2 * it)
        desc0, desc1 = model.forward_one_layer(desc0, desc1, 2 * it + 1)
        if it not in VALID_ITS:
            continue
        pred_score = model.compute_score(model.compute_distance(desc0, desc1, it))
        i0, _, m0, _ = compute_matches(pred_score, match_ratio)
        if trace is not None:
            trace.append({'it': it, 'n0': desc0.shape[1], 'n1': desc1.shape[1], 'indices0': i0[0].clone(),
                          'mscores0': m0[0].clone(), 'keep0': keep0.clone(), 'keep1': keep1.clone()})
        if int((i0 > -1).sum()) < min_kpts:                       # eval/matching.py:63-66
            last_R = last_t = None
            continue
        ids0 = torch.where(i0[0] > -1)[0]
        ids1 = i0[0][ids0]
        if ids0.numel() == 0:
            continue
        ret = None
        if estimate_pose is not None:                             # eval/matching.py:84-87
            ret = estimate_pose(kpts0=pts0[keep0[ids0].numpy()], kpts1=pts1[keep1[ids1].numpy()], K0=data.get('K0'),
                                K1=data.get('K1'), norm_thresh=error_th, method=method)
        if ret is not None:                                       # eval/matching.py:89-96,221-230
            _, R, t, inl = ret
            inl = torch.as_tensor(inl, dtype=torch.bool)
            inlier_ratio = float(inl.sum()) / ids0.numel()
        else:
            R = t = None
            inl = torch.zeros(ids0.numel(), dtype=torch.bool)
            inlier_ratio = 0.0
        # eval/matching.py:97-107 (it >= 3 at every scored iteration, so the it >= 1 branch always applies)
        diff_R = _angle_mat(last_R, R) if last_R is not None and R is not None else math.inf
        diff_t = _angle_vec(last_t, t) if last_t is not None and t is not None else math.inf
        pose_diff = max(diff_R, diff_t)
        last_R, last_t = R, t
        if uncertainty:                                           # eval/matching.py:243-257
            th = 0.2 * inlier_ratio if (with_uncertainty and inlier_ratio != 0) else 0.2
            sel0, sel1 = model.pool(pred_score, model.self_prob0, model.cross_prob0, model.self_prob1,
                                    model.cross_prob1, mscore_th=th, uncertainty_ratio=1.0)
        if stop_pose is not None and pose_diff <= stop_pose:      # eval/matching.py:110-117,260-268
            out_i = torch.full_like(i0[0], -1)
            out_i[ids0[inl]] = ids1[inl]
            return {'indices0': out_i, 'mscores0': m0[0], 'n_iter': it + 1, 'keep0': keep0, 'keep1': keep1,
                    'R': R, 't': t}
    i0, _, m0, _ = compute_matches(pred_score, 0.2)               # eval/matching.py:119,271
    return {'indices0': i0[0], 'mscores0': m0[0], 'n_iter': nI, 'keep0': keep0, 'keep1': keep1, 'R': None, 't': None}
