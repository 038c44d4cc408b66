"""BASELINE configs[1] (GM, N=1024, L=9, T=100, batch 1) a few times: the workload of the batch-1 profile"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imp_release_amd as P
from imp_release_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': 100, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 9,
       'GNN_layers': ['self', 'cross'] * 9, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = P.GM(cfg).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.to('cuda')
pair = synthetic.make_correlated_pair(N, N, seed=5, batch=B)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
with torch.no_grad():
    for _ in range(12):
        m.produce_matches(d, p=0.2, only_last=True)
torch.cuda.synchronize()
