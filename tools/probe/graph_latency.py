"""Probe: eager vs hipGraph replay of the fused one-shot call at batch 1 (single-pair latency)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=9, sinkhorn_iterations=100)
m = make_hip_model('GM', cfg, synthetic.make_state_dict(cfg, 'GM', seed=1))
ctx = m._ensure_ctx()
for B, N in ((1, 1024), (1, 2048), (4, 2048)):
    pair = synthetic.make_correlated_pair(N, N, seed=3, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    args = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'], 640.0, 480.0,
            float(m.bin_score), 100, True, 0.2)
    out = ctx.match_pair(*args)
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = timeit(lambda: ctx.match_pair(*args, out=out))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.match_pair(*args, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.match_pair(*args, out=out)
        torch.cuda.synchronize()
        replay = timeit(g.replay)
    print('B=%d N=%d: eager %.3f ms   graph replay %.3f ms' % (B, N, eager, replay))
