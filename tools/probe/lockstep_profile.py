"""Where a lock-step group's time goes (round 4): cProfile of eval_loop.run_pairs_sharded(lockstep=4, workers=1) on the harder evaluation set,
then wall-clock rates for a few (lockstep, workers) settings.   python tools/probe/lockstep_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model          # noqa: E402
from imp_release_amd import eval_loop, pose as gpose, synthetic   # noqa: E402

dev = torch.device('cuda', 0)
cfg = eval_config()
sd = synthetic.make_state_dict(cfg, 'DGNNS', seed=0, bin_score=synthetic.MATCHING_BIN_SCORE, style='matching')
m = make_hip_model('DGNNS', cfg, sd)
host_pairs = [synthetic.make_hard_two_view_pair(seed=7000 + i) for i in range(48)]
UP = ('keypoints0', 'keypoints1', 'scores0', 'scores1', 'descriptors0', 'descriptors1')
pinned = [{k: torch.from_numpy(pr[k]).pin_memory() for k in UP} for pr in host_pairs]


def provider(pid):
    pr = host_pairs[pid % len(host_pairs)]
    dd = {k: pinned[pid % len(host_pairs)][k].to(dev, non_blocking=True) for k in UP}
    dd['image0'] = dd['image1'] = torch.empty(pr['image_shape'], device='meta')
    dd['pts0_cpu'], dd['pts1_cpu'] = pr['keypoints0'][0], pr['keypoints1'][0]
    dd.update({k: pr[k] for k in ('K0', 'K1', 'T_0to1', 'E')})
    return dd


kw = dict(estimate_pose=gpose.estimate_pose)
eval_loop.run_pairs_sharded(m, provider, 16, lockstep=4, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
eval_loop.run_pairs_sharded(m, provider, 96, lockstep=4, **kw)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats('tottime').print_stats(32)
for ls, wk in ((1, 1), (1, 3), (4, 1), (4, 2), (4, 3), (2, 3), (8, 1)):
    reps = eval_loop.replicate(m, wk)
    try:
        eval_loop.run_pairs_sharded(m, provider, 24, lockstep=ls, workers=wk, replicas=reps, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eval_loop.run_pairs_sharded(m, provider, 480, lockstep=ls, workers=wk, replicas=reps, **kw)
        torch.cuda.synchronize()
        print(f'lockstep {ls} x workers {wk}: {480 / (time.perf_counter() - t0):.1f} pairs/s', flush=True)
    except Exception as e:
        print(f'lockstep {ls} x workers {wk}: {type(e).__name__}: {str(e)[:200]}', flush=True)
