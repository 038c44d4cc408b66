"""same-box A/B of the attention launch (HIP events on the launch stream, `reps` back-to-back launches, B x N of the bench):
    IMP_HIP_LIB=<variant> python tools/probe/attn_ab.py [B N reps rounds]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
B, N, reps, rounds = (int(v) for v in (sys.argv[1:5] + ['4', '2048', '10', '6'][len(sys.argv) - 1:]))
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
m.produce_matches(d, p=0.2, only_last=True)
print(os.environ.get('IMP_HIP_LIB', 'product library'), ' '.join('%.2f' % (ctx.time_attention(B, N, reps) * 1e3) for _ in range(rounds)), 'us')
