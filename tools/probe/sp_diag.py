#!/usr/bin/env python3
"""Stage-by-stage deviations of the SuperPoint HIP path against oracle/superpoint_oracle.py (test infrastructure; never asserts).
    python tools/probe/sp_diag.py [H W]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imp_release_amd import synthetic                      # noqa: E402
from imp_release_amd.superpoint import SuperPoint          # noqa: E402
from oracle import superpoint_oracle as spo                # noqa: E402

torch.set_num_threads(16)
dev = torch.device('cuda:0')
sd = synthetic.make_superpoint_state_dict(seed=0)
sp = SuperPoint({'state_dict': sd, 'max_keypoints': -1}, device=dev)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


print('--- single convolutions (random NCHW inputs) vs torch conv2d')
g = torch.Generator().manual_seed(0)
specs = [(0, 'conv1a', 1, 37, 53, True, False), (1, 'conv1b', 64, 24, 40, True, True), (1, 'conv1b', 64, 17, 23, True, False),
         (2, 'conv2a', 64, 16, 16, True, False), (4, 'conv3a', 64, 20, 28, True, False), (5, 'conv3b', 128, 22, 30, True, True),
         (7, 'conv4b', 128, 9, 11, True, False), (8, 'heads', 128, 15, 20, True, False), (9, 'convDb', 256, 15, 20, False, False)]
for layer, name, cin, H, W, relu, pool in specs:
    x = torch.randn(2, cin, H, W, generator=g)
    if layer == 8:
        wt = torch.cat([tsd['convPa.weight'], tsd['convDa.weight']]); bs = torch.cat([tsd['convPa.bias'], tsd['convDa.bias']])
    else:
        wt, bs = tsd[name + '.weight'], tsd[name + '.bias']
    ref = F.conv2d(x.double(), wt.double(), bs.double(), padding=wt.shape[-1] // 2)
    if relu:
        ref = torch.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    try:
        out = sp.op_conv(layer, x.to(dev), relu=relu, pool=pool).cpu().double()
        print(f'{name:8s} cin={cin:3d} {H}x{W} pool={int(pool)}: shape {tuple(out.shape)} max rel err {rel(out, ref):.2e}')
    except Exception as e:           # noqa: BLE001
        print(f'{name}: FAILED {e}')

shapes = [(96, 128), (100, 150), (240, 320)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for H, W in shapes:
    print(f'--- full network {H}x{W}')
    img = torch.from_numpy(synthetic.make_image(H, W, seed=3))
    with torch.no_grad():
        x = spo.encoder(sd, img)
        ds = spo.dense_scores(sd, x)
        dd = spo.dense_descriptors(sd, x)
        nm = spo.simple_nms(ds, 4)
        o = spo.forward(sd, img, align_corners=False)
    s_gpu, d_gpu = sp.extract({'image': img.to(dev)})
    n_gpu, _ = sp.extract({'image': img.to(dev)}, nms=True)
    print('dense scores  max abs err', float((s_gpu.cpu() - ds).abs().max()), ' max', float(ds.max()))
    print('dense desc    max abs err', float((d_gpu.cpu() - dd).abs().max()))
    # NMS kernel on its own: feed it the GPU's dense scores through the oracle
    nm_on_gpu_scores = spo.simple_nms(s_gpu.cpu(), 4)
    print('nms map (same input) mismatching pixels', int((n_gpu.cpu() != nm_on_gpu_scores).sum()), ' vs oracle end-to-end',
          int(((n_gpu.cpu() > 0) != (nm > 0)).sum()))
    for mk in (-1, 100):
        sp.config['max_keypoints'] = mk
        sp.config['align_corners'] = False
        out = sp({'image': img.to(dev)})
        oo = spo.forward(sd, img, max_keypoints=mk, align_corners=False)
        kg, ko = out['keypoints'][0].cpu(), oo['keypoints'][0]
        same = kg.shape == ko.shape and bool(torch.equal(kg, ko))
        print(f'forward mk={mk}: n gpu {len(kg)} oracle {len(ko)} identical {same}', end='')
        if same:
            print('  dscore', float((out['scores'][0].cpu() - oo['scores'][0]).abs().max()),
                  ' ddesc', float((out['descriptors'][0].cpu() - oo['descriptors'][0]).abs().max()))
        else:
            a, b = set(map(tuple, kg.int().tolist())), set(map(tuple, ko.int().tolist()))
            print('  symmetric difference', len(a ^ b))
    sp.config['max_keypoints'] = -1

print('--- timing 480x640, top-1024')
img = torch.from_numpy(synthetic.make_image(480, 640, seed=5)).to(dev)
sp.config['max_keypoints'] = 1024
for _ in range(3):
    sp({'image': img})
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    sp({'image': img})
torch.cuda.synchronize()
print('ms per image', (time.perf_counter() - t0) / 20 * 1e3)
