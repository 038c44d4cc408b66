"""Probe: aggregate throughput of K independent matcher instances, each on its own stream + host thread (batch 4, N=2048)."""
import sys, threading, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
sd = synthetic.make_state_dict(cfg, 'GM', seed=1)
B, N, STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 2048, 20
def mk(seed):
    m = make_hip_model('GM', cfg, sd)
    pair = synthetic.make_correlated_pair(N, N, seed=seed, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    return m, d
for K in (1, 2, 3):
    inst = [mk(10 + i) for i in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    def work(i, steps):
        m, d = inst[i]
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(steps):
                m.produce_matches(d, p=0.2, only_last=True)
            streams[i].synchronize()
    for phase, steps in (('warm', 3), ('timed', STEPS)):
        th = [threading.Thread(target=work, args=(i, steps)) for i in range(K)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%d instance(s) x batch %d: %.1f pairs/s aggregate (%.2f ms per batch-step)' % (K, B, K * B * STEPS / dt, dt / (K * STEPS) * 1e3))
