"""Phase profile of the resident Sinkhorn kernel (IMP_OPTIONS=probe_prof=1 prints cycles per iteration of workgroup 0 per phase):
    IMP_OPTIONS=probe_prof=1 python tools/probe/sk_prof.py [B N]..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model          # noqa: E402
from imp_release_amd import synthetic                    # noqa: E402

a = [int(x) for x in sys.argv[1:]]
shapes = list(zip(a[0::2], a[1::2])) or [(1, 1024), (1, 512), (4, 1024), (4, 2048), (1, 2048)]
cfg = eval_config(n_layers=1)
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd)
ctx = m._ensure_ctx()
for B, N in shapes:
    pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    m.produce_matches(d, p=0.2, only_last=True)
    print(B, N, 'sinkhorn us/iter', round(ctx.time_sinkhorn(B, N, 50) * 1e3, 2), ctx.resident_status(), flush=True)
