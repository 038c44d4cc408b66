"""debug: does the first call on a fresh context differ from the second? (run with IMP_POISON_WORKSPACE=1)"""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=3, sinkhorn_iterations=20)
sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
for n0, n1, B in ((1024, 1000, 2), (1024, 1024, 2), (700, 640, 2), (1024, 1000, 1)):
    m = make_hip_model('GM', cfg, sd)
    pair = synthetic.make_correlated_pair(n0, n1, seed=60, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    outs = []
    for rep in range(3):
        o = m.produce_matches(d, p=0.2, only_last=True)
        outs.append((o['indices0'][-1].clone(), o['mscores0'][-1].clone(), o['scores'][-1].clone()))
    for rep in (1, 2):
        di = (outs[rep][0] != outs[0][0]).sum().item()
        dm = (outs[rep][1] - outs[0][1]).abs().max().item()
        ds = (outs[rep][2] - outs[0][2]).abs().max().item()
        print(n0, n1, B, 'call', rep, 'vs 0: idx diff', di, 'ms diff', dm, 'score diff', ds, 'nan', torch.isnan(outs[0][2]).sum().item())
    # step API: layer by layer on a fresh model vs the used one
    m2 = make_hip_model('GM', cfg, sd)
    ctxs = [m2._ensure_ctx(), m._ensure_ctx()]
    res = []
    for ctx in ctxs:
        nk0 = ctx.normalize_keypoints(d['keypoints0'], 640, 480); nk1 = ctx.normalize_keypoints(d['keypoints1'], 640, 480)
        e0, e1 = ctx.encode_keypoints(nk0, d['scores0'], nk1, d['scores1'], d['descriptors0'], d['descriptors1'])
        r = [e0.clone(), e1.clone()]
        for li in range(6):
            e0, e1 = ctx.forward_layer(li, e0, e1)
            r += [e0.clone(), e1.clone()]
        dist = ctx.compute_distance(2, e0, e1); r.append(dist.clone())
        sc = ctx.compute_score(dist, 1.0, 20, True); r.append(sc.clone())
        res.append(r)
    print('  stage diffs fresh vs used:', [f'{(a - b).abs().max().item():.1e}' for a, b in zip(*res)])
