"""cProfile of the host side of one-shot produce_matches calls (GM, N = 2048, batch 4) - where the time between two calls goes (round 5)"""
import cProfile, pstats, io, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imp_release_amd as P
from imp_release_amd import synthetic
cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': 100, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 9,
       'GNN_layers': ['self', 'cross'] * 9, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
pairs = [synthetic.make_correlated_pair(2048, 2048, seed=100 + i) for i in range(4)]
d = {k: torch.from_numpy(np.concatenate([p[k] for p in pairs], 0)).cuda() for k in pairs[0] if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pairs[0]['image_shape'], device='cuda')
m = P.GM(dict(cfg, range_recovery=False)).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.cuda()
with torch.no_grad():
    for _ in range(5): m.produce_matches(d, p=0.2, only_last=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): m.produce_matches(d, p=0.2, only_last=True)
    host = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    print('host time per ASYNC call (enqueue only): %.3f ms' % host)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): m.produce_matches(d, p=0.2, only_last=True)
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:4000])
