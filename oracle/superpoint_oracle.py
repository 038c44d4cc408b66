"""CPU ORACLE for the SuperPoint front-end.   *** TEST INFRASTRUCTURE ONLY ***

torch-CPU fp32 functional restatement of nets/superpoint.py (every function cites the lines it follows), pinned by
tools/make_golden.py against the imported reference on seeded random weights (superpoint_v1.pth is absent here) ->
tests/golden/superpoint_*.npz, re-checked by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _conv(sd, name, x, pad):
    return F.conv2d(x, torch.as_tensor(sd[name + '.weight']), torch.as_tensor(sd[name + '.bias']), padding=pad)


def encoder(sd, image):
    """nets/superpoint.py:142-153 (shared VGG-style encoder, three 2x2 max pools): [B,1,H,W] -> [B,128,H/8,W/8]"""
    x = torch.relu(_conv(sd, 'conv1a', image, 1))
    x = torch.relu(_conv(sd, 'conv1b', x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = torch.relu(_conv(sd, 'conv2a', x, 1))
    x = torch.relu(_conv(sd, 'conv2b', x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = torch.relu(_conv(sd, 'conv3a', x, 1))
    x = torch.relu(_conv(sd, 'conv3b', x, 1))
    x = F.max_pool2d(x, 2, 2)
    x = torch.relu(_conv(sd, 'conv4a', x, 1))
    return torch.relu(_conv(sd, 'conv4b', x, 1))


def dense_scores(sd, x):
    """nets/superpoint.py:155-161: detector head, softmax over 65 bins, dustbin dropped, 8x8 pixel shuffle -> [B, H, W]"""
    s = _conv(sd, 'convPb', torch.relu(_conv(sd, 'convPa', x, 1)), 0)
    s = torch.softmax(s, 1)[:, :-1]
    b, _, h, w = s.shape
    s = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
    return s.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)


def dense_descriptors(sd, x):
    """nets/superpoint.py:163-166: descriptor head, L2-normalised over channels -> [B, D, H/8, W/8]"""
    return F.normalize(_conv(sd, 'convDb', torch.relu(_conv(sd, 'convDa', x, 1)), 0), p=2, dim=1)


def _window_max(t, radius):
    return F.max_pool2d(t, 2 * radius + 1, 1, radius)


def simple_nms(scores, radius):
    """nets/superpoint.py:49-63, restated: a pixel survives when it is the maximum of its (2 radius + 1)^2 window; twice more, pixels
    farther than `radius` from every survivor compete among themselves (the neighbourhoods of the survivors count as zero) and the
    window maxima among THEM survive as well.  Everything else becomes 0."""
    keep = scores == _window_max(scores, radius)
    for _ in range(2):
        taken = _window_max(keep.to(scores.dtype), radius) > 0            # within `radius` of a survivor
        rest = scores.masked_fill(taken, 0.0)
        keep = keep | ((rest == _window_max(rest, radius)) & ~taken)
    return scores.masked_fill(~keep, 0.0)


def sample_descriptors(kpts_xy, desc, s=8, align_corners=True):
    """nets/superpoint.py:82-94: bilinear sampling of the coarse descriptor map [1, D, h, w] at pixel keypoints [1, N, 2] (x, y), then L2
    normalisation.  A pixel coordinate p maps to the sampling coordinate 2 (p - s/2 + 1/2) / (n s - s/2 - 1/2) - 1 (n = w for x, h for y;
    same operation order as the reference, so the same bits).  align_corners: the reference passes True only when
    int(torch.__version__[2]) > 2 (:89), i.e. torch 1.3-1.9; on torch 2.x grid_sample's default False applies."""
    n_img, depth, h, w = desc.shape
    span = torch.tensor([w * s - s / 2 - 0.5, h * s - s / 2 - 0.5]).to(kpts_xy)
    grid = (((kpts_xy - s / 2 + 0.5) / span[None]).to(desc.dtype) * 2 - 1).view(n_img, 1, -1, 2)
    picked = F.grid_sample(desc, grid, mode='bilinear', align_corners=align_corners)
    return F.normalize(picked.reshape(n_img, depth, -1), p=2, dim=1)


def forward(sd, image, nms_radius=4, keypoint_threshold=0.0025, max_keypoints=-1, remove_borders=4, align_corners=True):
    """nets/superpoint.py:170-232 -> dict of per-image lists: keypoints [N,2] (x, y) float, scores [N], descriptors [D, N]"""
    x = encoder(sd, image)
    scores = simple_nms(dense_scores(sd, x), nms_radius)
    desc = dense_descriptors(sd, x)
    H, W = scores.shape[1:]
    out = {'keypoints': [], 'scores': [], 'descriptors': []}
    for b in range(image.shape[0]):
        kp = torch.nonzero(scores[b] > keypoint_threshold)                       # (row, col), row-major order
        sc = scores[b][tuple(kp.t())]
        m = (kp[:, 0] >= remove_borders) & (kp[:, 0] < H - remove_borders) & (kp[:, 1] >= remove_borders) & (kp[:, 1] < W - remove_borders)
        kp, sc = kp[m], sc[m]
        if max_keypoints >= 0 and max_keypoints < len(kp):
            sc, idx = torch.topk(sc, max_keypoints, dim=0)
            kp = kp[idx]
        kp = torch.flip(kp, [1]).float()
        out['keypoints'].append(kp)
        out['scores'].append(sc)
        out['descriptors'].append(sample_descriptors(kp[None], desc[b:b + 1], 8, align_corners)[0])
    return out
