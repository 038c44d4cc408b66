"""attention launch time at small grids for the lock-step kernels with 2 / 4 / 8 waves (64 / 128 / 256 queries per workgroup)
and the ping-pong kernel; run once per setting: IMP_ATTN_WAVES=2|4|8 or IMP_ATTN_VARIANT=2"""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
tag = f"waves={os.environ.get('IMP_ATTN_WAVES', '-')} variant={os.environ.get('IMP_ATTN_VARIANT', '-')}"
row = []
for B, N in ((1, 512), (1, 1024), (1, 2048), (2, 1024), (1, 4096), (4, 2048)):
    pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    m.produce_matches(d, p=0.2, only_last=True)
    row.append(f'B{B} N{N}: {ctx.time_attention(B, N, 20) * 1e3:6.1f}us')
print(tag, ' | '.join(row))
