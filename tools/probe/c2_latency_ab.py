"""configs[1] latency (GM, N = 1024, batch 1, L = 9, T = 100) under the settings of the in-call range recovery (round 5)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imp_release_amd as P
from imp_release_amd import synthetic
cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': 100, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 9,
       'GNN_layers': ['self', 'cross'] * 9, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
pair = synthetic.make_correlated_pair(1024, 1024, seed=5)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
for rr in (True, False):
    m = P.GM(dict(cfg, range_recovery=rr)).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda()
    with torch.no_grad():
        for _ in range(5): m.produce_matches(d, p=0.2, only_last=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): m.produce_matches(d, p=0.2, only_last=True)
        torch.cuda.synchronize()
    print('range_recovery %s  IMP_RANGE_SPIN=%s: %.3f ms per call' % (rr, os.environ.get('IMP_RANGE_SPIN', '1'), (time.perf_counter() - t0) / 40 * 1e3), flush=True)
