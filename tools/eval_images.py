#!/usr/bin/env python3
"""Image pair in -> matches out, everything on one GPU: 2 x SuperPoint (nets/superpoint.py) + GM one-shot matcher (nets/gm.py
produce_matches) per pair, K pairs in flight (pipeline.StepPipeline: K replicas of both models, one host thread and stream each).

    python tools/eval_images.py [--pairs 64] [--workers 3] [--height 480 --width 640] [--kpts 1024] [--iters 9] [--sinkhorn 100]

Synthetic images and seeded random weights (no checkpoints offline): this measures the pipeline, not matching quality.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imp_release_amd as P                                   # noqa: E402
from imp_release_amd import synthetic                         # noqa: E402
from imp_release_amd.pipeline import StepPipeline             # noqa: E402
from imp_release_amd.superpoint import SuperPoint             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=64)
    ap.add_argument('--workers', type=int, default=3)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--kpts', type=int, default=1024)
    ap.add_argument('--iters', type=int, default=9)
    ap.add_argument('--sinkhorn', type=int, default=100)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': a.sinkhorn, 'match_threshold': 0.2, 'with_sinkhorn': True,
           'n_layers': a.iters, 'GNN_layers': ['self', 'cross'] * a.iters, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
    msd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(cfg, 'GM', seed=0).items()}
    ssd = synthetic.make_superpoint_state_dict(seed=1)
    imgs = [torch.from_numpy(synthetic.make_image(a.height, a.width, seed=s)).to(dev) for s in range(8)]

    def replica(i):
        sp = SuperPoint({'state_dict': ssd, 'max_keypoints': a.kpts}, device=dev)
        gm = P.GM(cfg).eval()
        gm.load_state_dict(msd, strict=True)
        gm = gm.to(dev)
        state = {'n': i}

        def step():
            k = state['n']
            state['n'] += a.workers
            d = {}
            pair = torch.cat([imgs[(2 * k) % len(imgs)], imgs[(2 * k + 1) % len(imgs)]])       # both images in ONE front-end call
            o = sp({'image': pair})
            for j in (0, 1):
                d[f'keypoints{j}'] = o['keypoints'][j][None]
                d[f'scores{j}'] = o['scores'][j][None]
                d[f'descriptors{j}'] = o['descriptors'][j].t()[None].contiguous()
                d[f'image{j}'] = pair[j:j + 1]
            out = gm.produce_matches(d, p=0.2, only_last=True)
            return out['indices0'][-1], out['mscores0'][-1]
        return step

    pipe = StepPipeline([replica(i) for i in range(a.workers)], n_total=1, device=dev, exchange=lambda i0, m0: (i0, m0))
    pipe.run(2 * a.workers)                                   # warm-up (workspace growth, first-call allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i0, m0 = pipe.run(a.pairs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{a.pairs} image pairs {a.height}x{a.width}, top-{a.kpts} keypoints, GM L={a.iters} T={a.sinkhorn}, {a.workers} pairs in flight: '
          f'{a.pairs / dt:.1f} image pairs/s ({dt / a.pairs * 1e3:.2f} ms per pair); last pair: {int((i0 >= 0).sum())} matches of {i0.shape[1]}')


if __name__ == '__main__':
    main()
