"""Probe: the EIMP sliced loop (BASELINE config 4 shape) a few times, for rocprofv3 kernel statistics."""
import sys, time, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic, matching
cfg = eval_config()
sd = synthetic.make_state_dict(cfg, 'AdaGMN', seed=0, bin_score=5.0)
m = make_hip_model('AdaGMN', cfg, sd)
pair = synthetic.make_correlated_pair(4096, 4000, seed=5)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
d['pts0_cpu'], d['pts1_cpu'] = pair['keypoints0'][0], pair['keypoints1'][0]
with torch.no_grad():
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = matching.matching_iterative_uncertainty(d, m, 15, 0.1, 25, 1.0, {'pose': 1.5})
        torch.cuda.synchronize(); print('loop %.2f ms' % ((time.perf_counter() - t0) * 1e3))
