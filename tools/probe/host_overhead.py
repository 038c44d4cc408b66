"""Probe: host enqueue time vs GPU time per step for ctx.match_pair and GM.produce_matches (B=4, N=2048)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
m = make_hip_model('GM', cfg, synthetic.make_state_dict(cfg, 'GM', seed=1))
ctx = m._ensure_ctx()
B, N = 4, 2048
pair = synthetic.make_correlated_pair(N, N, seed=3, batch=B)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'], 640.0, 480.0,
        float(m.bin_score.detach()), 100, True, 0.2)
def bench(fn, n=20):
    with torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
for name, fn in (('match_pair', lambda: ctx.match_pair(*args)),
                 ('match_pair +scores +side1', lambda: ctx.match_pair(*args, want_scores=True, want_side1=True)),
                 ('produce_matches(only_last)', lambda: m.produce_matches(d, p=0.2, only_last=True))):
    h, t = bench(fn)
    print('%-28s host enqueue %.3f ms/step   wall %.3f ms/step' % (name, h, t))
