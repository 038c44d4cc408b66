"""Layer GEMMs: the weight-fragment kernel (gemm_wf.hip) against gemm_f32.hip on the same shapes, HIP events, 20 launches each.
    python tools/probe/gemm_wf_time.py [B N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model          # noqa: E402
from imp_release_amd import synthetic                    # noqa: E402

shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(4, 2048), (1, 1024), (1, 2048), (2, 2048), (3, 2048), (8, 2048)]
cfg = eval_config(n_layers=1)
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd)
ctx = m._ensure_ctx()
for B, N in shapes:
    pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    m.produce_matches(d, p=0.2, only_last=True)
    D = 256
    flops = {'QKV': 2 * 2 * B * N * D * 3 * D, 'MLP0': 2 * 2 * B * N * 2 * D * 2 * D, 'MLP3': 2 * 2 * B * N * 2 * D * D}
    for which, name in enumerate(('QKV', 'MLP0', 'MLP3')):
        t_old = ctx.time_layer_gemm(B, N, which, -1) * 1e3
        t_new = ctx.time_layer_gemm(B, N, which, -2) * 1e3
        probes = '  '.join(f'{tag} {ctx.time_layer_gemm(B, N, which, d) * 1e3:5.1f}' for d, tag in ((-3, 'no-stores'), (-4, 'no-epilogue'), (-6, 'no-bias/residual')))
        print(f'B={B} N={N} {name:5s}: gemm_f32 {t_old:6.1f} us   gemm_wf {t_new:6.1f} us   ({t_old / t_new:4.2f}x, '
              f'{3 * flops[name] / t_new / 1e6 / 2500:.3f} of 2.5 PF executed)   probes: {probes}')
    # the chained launch: conv 3 + the next layer's q|k|v projection on the same 64-row tile
    t_c = ctx.time_layer_gemm(B, N, 3, -2) * 1e3
    t_sep = ctx.time_layer_gemm(B, N, 2, -2) * 1e3 + min(ctx.time_layer_gemm(B, N, 0, -1), ctx.time_layer_gemm(B, N, 0, -2)) * 1e3
    fl = flops['MLP3'] + flops['QKV']
    print(f'B={B} N={N} MLP3+QKV chained {t_c:6.1f} us  (separately {t_sep:6.1f} us, {t_sep / t_c:4.2f}x, {3 * fl / t_c / 1e6 / 2500:.3f} of 2.5 PF executed)'
          f'   no-stores {ctx.time_layer_gemm(B, N, 3, -3) * 1e3:5.1f}')
