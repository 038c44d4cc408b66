"""which replicas of a three-in-flight pipeline should take the fused layer launch?  per-replica option wf_fused (0 never, 1 by the count rule, 2 always):
    python tools/probe/fused_per_replica.py            -> pairs/s for several assignments, interleaved, two rounds"""
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch  # noqa: E402
from helpers import eval_config, make_hip_model  # noqa: E402
from imp_release_amd import eval_loop, pipeline, synthetic  # noqa: E402

DEV = torch.device('cuda', 0)
cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
pair = synthetic.make_correlated_pair(2048, 2048, seed=7, batch=4)
data = {k: torch.from_numpy(v).to(DEV) for k, v in pair.items() if k != 'image_shape'}
data['image0'] = data['image1'] = torch.zeros(pair['image_shape'], device=DEV)


def build(assign):
    reps = []
    for v in assign:
        m = make_hip_model('GM', cfg, sd)
        m._ensure_ctx().option('wf_fused', v)
        reps.append(m)
    return reps


def make_step(mm):
    def fn():
        out = mm.produce_matches(data, p=0.2, only_last=True)
        return out['indices0'][-1], out['mscores0'][-1]
    return fn


def rate(reps, n=40):
    pp = pipeline.StepPipeline([make_step(r) for r in reps], 4, device=DEV)
    pp.run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pp.run(n)
    torch.cuda.synchronize()
    return 4 * n / (time.perf_counter() - t0)


ASSIGN = [(1, 1, 1), (2, 0, 0), (2, 2, 0), (2, 1, 1), (2, 2, 2), (0, 0, 0), (1, 1), (2, 0), (2, 2)]
models = {a: build(a) for a in ASSIGN}
with torch.no_grad():
    for rnd in range(2):
        for a in ASSIGN:
            print(f'wf_fused per replica {a}: {rate(models[a]):.1f} pairs/s', flush=True)
