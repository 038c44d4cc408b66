#!/usr/bin/env python3
"""SuperPoint front-end timing: python tools/probe/sp_time.py [H W [max_keypoints [iters [batch]]]]  (run under rocprofv3 for per-kernel times)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imp_release_amd import synthetic                      # noqa: E402
from imp_release_amd.superpoint import SuperPoint          # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 480
W = int(sys.argv[2]) if len(sys.argv) > 2 else 640
mk = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device('cuda:0')
sp = SuperPoint({'state_dict': synthetic.make_superpoint_state_dict(seed=1), 'max_keypoints': mk}, device=dev)
img = torch.from_numpy(synthetic.make_image(H, W, seed=5, batch=batch)).to(dev)
for _ in range(5):
    out = sp({'image': img})
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    out = sp({'image': img})
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f'{H}x{W} batch {batch} top-{mk}: {dt * 1e3:.3f} ms per call, {batch / dt:.1f} images/s, n = {[len(k) for k in out["keypoints"]]}')
