"""Layer GEMMs at the bench geometry: the two launches of round 3 against the fused launch of round 4 (HIP events, 20 launches each;
with IMP_OPTIONS=probe_prof=1 and a -DWF_PROFILE build of the library: phase cycle stamps of the fused kernel).   python tools/probe/fused_time.py [B N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model          # noqa: E402
from imp_release_amd import synthetic                    # noqa: E402

B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 2048)
cfg = eval_config(n_layers=1)
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd)
ctx = m._ensure_ctx()
pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
m.produce_matches(d, p=0.2, only_last=True)
for rep in range(3):
    t = {w: ctx.time_layer_gemm(B, N, w, -2) * 1e3 for w in (1, 3, 4)}
    print(f'B={B} N={N}: MLP0 {t[1]:.1f}  MLP3+QKV chained {t[3]:.1f}  | two launches {t[1] + t[3]:.1f} us  FUSED {t[4]:.1f} us', flush=True)
