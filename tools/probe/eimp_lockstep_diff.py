"""solo EIMP loop vs groups of 4 in lock step on the harder synthetic set: where (pair, iteration) do the two first differ?"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from imp_release_amd import synthetic, matching, pose as gpose
import imp_release_amd as P

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
cfg = {'descriptor_dim': 256, 'sinkhorn_iterations': 20, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 15,
       'GNN_layers': ['self', 'cross'] * 15, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}
sd = synthetic.make_state_dict(cfg, 'AdaGMN', seed=0, bin_score=synthetic.MATCHING_BIN_SCORE, style='matching')
m = P.AdaGMN(cfg).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.to(dev)
UP = ('keypoints0', 'keypoints1', 'scores0', 'scores1', 'descriptors0', 'descriptors1')


def data_of(pid):
    pair = synthetic.make_hard_two_view_pair(seed=1000 + pid)
    d = {k: torch.from_numpy(pair[k]).to(dev) for k in UP}
    d['image0'] = d['image1'] = torch.empty(pair['image_shape'], device='meta')
    d['pts0_cpu'] = pair['keypoints0'][0]; d['pts1_cpu'] = pair['keypoints1'][0]
    d.update({k: pair[k] for k in ('K0', 'K1', 'T_0to1', 'E')})
    return d


bad = 0
with torch.no_grad():
    for g0 in range(0, n_pairs, 4):
        datas = [data_of(p) for p in range(g0, min(g0 + 4, n_pairs))]
        st, lt = [[] for _ in datas], [[] for _ in datas]
        solo = [matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5}, estimate_pose=gpose.estimate_pose, trace=st[i])
                for i, d in enumerate(datas)]
        lock = matching.matching_iterative_uncertainty_lockstep(datas, m, 15, 0.1, 25, 1.0, {'pose': 1.5}, estimate_pose=gpose.estimate_pose, traces=lt)
        for i, (a, c) in enumerate(zip(solo, lock)):
            same = a[8] == c[8] and np.array_equal(a[4], c[4]) and np.array_equal(a[0], c[0])
            if same:
                continue
            bad += 1
            print(f'pair {g0 + i}: n_iter {a[8]} vs {c[8]}; sizes {datas[i]["keypoints0"].shape[1]}/{datas[i]["keypoints1"].shape[1]}')
            for k, (x, y) in enumerate(zip(st[i], lt[i])):
                di = int((x['indices0'] != y['indices0']).sum()) if x['indices0'].shape == y['indices0'].shape else -1
                dm = float(np.abs(x['mscores0'] - y['mscores0']).max()) if x['mscores0'].shape == y['mscores0'].shape else -1
                print(f'   it {x["it"]}: n {x["n0"]}/{x["n1"]} vs {y["n0"]}/{y["n1"]}  index diffs {di}  max |dms| {dm:.3g}  matches {int((x["indices0"] >= 0).sum())}')
                if di != 0:
                    break
print('pairs that differ:', bad, 'of', n_pairs)
