#!/usr/bin/env python3
"""Timing of the other BASELINE configs on one GPU (not the bench metric): C2 (N=1024, batch 1), C1-like DGNNS,
C4 EIMP sliced loop from N=4096.  Prints GPU time per call and host-side launch rate."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imp_release_amd as P
from imp_release_amd import synthetic, matching

dev = 'cuda'
def cfg_of(n_layers, T):
    return {'descriptor_dim': 256, 'sinkhorn_iterations': T, 'match_threshold': 0.2, 'with_sinkhorn': True,
            'n_layers': n_layers, 'GNN_layers': ['self', 'cross'] * n_layers, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
def model_of(name, cfg, **kw):
    sd = synthetic.make_state_dict(cfg, name, seed=0, **kw)
    m = getattr(P, name)(cfg).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to(dev)
def data_of(n0, n1, B=1, seed=5):
    pair = synthetic.make_correlated_pair(n0, n1, seed=seed, batch=B)
    d = {k: torch.from_numpy(v).to(dev) for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device=dev)
    return d
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

with torch.no_grad():
    for N, B in ((1024, 1), (2048, 1), (2048, 4), (512, 1)):
        m = model_of('GM', cfg_of(9, 100)); d = data_of(N, N, B)
        ms = timeit(lambda: m.produce_matches(d, p=0.2, only_last=True))
        # same with a captured HIP graph of the fused call
        ctx = m._ensure_ctx()
        out = ctx.match_pair(d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'],
                             640., 480., 1.0, 100, True, 0.2)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                ctx.match_pair(d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'],
                               640., 480., 1.0, 100, True, 0.2, out=out)
            gms = timeit(lambda: g.replay())
        except Exception as e:
            gms = float('nan'); print('graph capture failed:', repr(e)[:200])
        print(f'GM L=9 T=100 N={N} B={B}: eager {ms:.3f} ms/call = {B/ms*1e3:.1f} pairs/s ; hipGraph replay {gms:.3f} ms = {B/gms*1e3:.1f} pairs/s')
    m = model_of('DGNNS', cfg_of(15, 20)); d = data_of(512, 519)
    ms = timeit(lambda: m.produce_matches(d, p=0.2, only_last=True))
    print(f'DGNNS L=15 T=20 N=512/519 B=1 (C1 analogue): {ms:.3f} ms/pair')
    m = model_of('AdaGMN', cfg_of(15, 20), bin_score=5.0); d = data_of(4096, 4000)
    d['pts0_cpu'] = d['keypoints0'][0].cpu().numpy(); d['pts1_cpu'] = d['keypoints1'][0].cpu().numpy()
    tr = []
    ms = timeit(lambda: matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5}, trace=tr), reps=5, warm=1)
    print(f'EIMP sliced loop N=4096/4000 (C4), 15 iters, 7 score+pool steps, pose stubbed: {ms:.2f} ms/pair; trajectory', [(t['n0'], t['n1']) for t in tr[-7:]])
