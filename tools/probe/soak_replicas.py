"""Soak of what bench.py does around its pipelines: build replicas, run a few steps with 1 / 2 / 3 in flight, drop them - for SECONDS seconds - and count the waiting
launches (chip-resident Sinkhorn, fused layer MLP) that timed out.  usage: python tools/probe/soak_replicas.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import imp_release_amd as P                                     # noqa: E402
from imp_release_amd import _lib, eval_loop, pipeline, synthetic  # noqa: E402
import bench                                                    # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
cfg = bench.eval_config(9, 100)
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
model = P.GM(dict(cfg, precision='f16x3')).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model = model.to(dev)
B, N = 4, 2048
pairs = [synthetic.make_correlated_pair(N, N, seed=100 + pid) for pid in range(B)]
data = {k: torch.from_numpy(np.concatenate([p[k] for p in pairs], 0)).to(dev) for k in pairs[0] if k != 'image_shape'}
data['image0'] = data['image1'] = torch.zeros(pairs[0]['image_shape'], device=dev)


def make_step(m):
    def step_fn():
        out = m.produce_matches(data, p=0.2, only_last=True)
        return out['indices0'][-1], out['mscores0'][-1]
    return step_fn


t_end = time.time() + secs
cycles = steps = voided = raised = 0
ref = None
while time.time() < t_end:
    for k in (1, 2, 3):
        reps = [model] if k == 1 else eval_loop.replicate(model, k)
        pp = pipeline.StepPipeline([make_step(m) for m in reps], B, device=dev, exchange_every=1)
        try:
            r = pp.run(12 + cycles % 5)
            torch.cuda.synchronize()
            steps += 12 + cycles % 5
            if ref is None:
                ref = (r[0].clone(), r[1].clone())
            elif not (torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1])):
                print(f'cycle {cycles} k={k}: RESULT DIFFERS from the first step ({int((r[0] != ref[0]).sum())} indices)', flush=True)
        except _lib.ResidentSinkhornTimeout as e:
            raised += 1
            print(f'cycle {cycles} k={k}: {str(e)[:160]}', flush=True)
            torch.cuda.synchronize()
        v = bench.voided_launches(reps)
        if v:
            voided += v
            print(f'cycle {cycles} k={k}: {v} voided launch(es) on these replicas', flush=True)
        del pp, reps
    cycles += 1
print(f'soak: {cycles} cycles, {steps} steps in {secs:.0f} s: {raised} calls raised, {voided} voided launches counted')
