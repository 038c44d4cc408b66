"""Where the time of the three layer GEMMs goes (HIP events, 20 back-to-back launches each): gemm_f32.hip and gemm_wf.hip with
parts of the latter switched off.  (The round-2 pre-split planes kernel this probe was written for is retired: tools/probe/gemm_planes.hip.)"""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 2048)
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
m.produce_matches(d, p=0.2, only_last=True)
names = ['QKV', 'MLP0', 'MLP3']
for which in range(3):
    row = [names[which]]
    for dbg, tag in ((-1, 'gemm_f32'), (-2, 'gemm_wf'), (-3, 'wf no-stores'), (-4, 'wf no-epilogue'), (-6, 'wf no-bias/residual loads')):
        row.append(f'{tag} {ctx.time_layer_gemm(B, N, which, dbg) * 1e3:6.1f}')
    print(' | '.join(row))
