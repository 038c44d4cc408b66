"""diagnostic (not collected): locate the mutual-NN flips of one soak case and show the score values around them"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
from oracle import imp_oracle as orc
case = {'model': 'DGNNS', 'n0': 1009, 'n1': 386, 'B': 3, 'L': 15, 'T': 20, 'D': 256, 'seed': 1058}
cfg = eval_config(n_layers=case['L'], sinkhorn_iterations=case['T'], descriptor_dim=case['D'])
sd = synthetic.make_state_dict(cfg, case['model'], seed=case['seed'])
pair = synthetic.make_correlated_pair(case['n0'], case['n1'], desc_dim=case['D'], seed=case['seed'], batch=case['B'])
data = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device='cuda')
m = make_hip_model(case['model'], cfg, sd, precision=os.environ.get('PREC'))
with torch.no_grad():
    out = m._run_iterations(data, 0.2, False, want_scores=True)
    ref = orc.MatcherOracle(cfg, sd, case['model']).produce_matches({k: v.cpu() for k, v in data.items()}, p=0.2)
for it in range(len(ref['mscores0'])):
    g_ms, r_ms = out['mscores0'][it].cpu().numpy(), ref['mscores0'][it].numpy()
    g_sc, r_sc = out['scores'][it].cpu().numpy(), ref['scores'][it].numpy()
    dis = np.argwhere((g_ms > 0) != (r_ms > 0))
    if len(dis) == 0:
        continue
    print('iteration', it, 'score matrix max abs diff (inner)', np.abs(g_sc - r_sc)[:, :-1, :-1].max())
    for b, i in dis:
        jg, jr = g_sc[b, i, :-1].argmax(), r_sc[b, i, :-1].argmax()
        cr, cg = r_sc[b, :-1, jr], g_sc[b, :-1, jr]
        print(f'  b={b} row {i}: got ms {g_ms[b, i]:.7f} ref ms {r_ms[b, i]:.7f}; row argmax got {jg} ref {jr}; col {jr}: '
              f'ref top2 rows {np.argsort(-cr)[:2]} vals {np.sort(cr)[-2:][::-1]}; got top2 rows {np.argsort(-cg)[:2]} vals {np.sort(cg)[-2:][::-1]}')
