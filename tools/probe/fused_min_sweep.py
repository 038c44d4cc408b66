"""one-shot latency by (B, N) with the fused layer launch forced on for every tile count (IMP_OPTIONS=wf_fused_min=1) or at its default threshold:
    IMP_OPTIONS=wf_fused_min=1 python tools/probe/fused_min_sweep.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=9, sinkhorn_iterations=100); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd)
out = []
for B, N in ((1, 512), (1, 1024), (2, 1024), (1, 2048), (4, 1024), (2, 2048), (1, 4096)):
    pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    for _ in range(3):
        m.produce_matches(d, p=0.2, only_last=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        m.produce_matches(d, p=0.2, only_last=True)
    torch.cuda.synchronize()
    out.append(f'B={B} N={N} ({2 * B * ((N + 63) // 64)} tiles): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms')
print('IMP_OPTIONS=' + os.environ.get('IMP_OPTIONS', 'default'), ' | '.join(out))
