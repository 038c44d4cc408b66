"""Probe: two independent Sinkhorn chains (2 pairs each) on two streams vs one chain of 4 pairs."""
import sys, threading, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1)
sd = synthetic.make_state_dict(cfg, model='GM', seed=1)
mods = [make_hip_model('GM', cfg, sd) for _ in range(4)]
ctxs = [m._ensure_ctx() for m in mods]
N, IT = 2048, 400
print('one chain  B=4: %.2f us/iteration' % (ctxs[0].time_sinkhorn(4, N, IT) * 2e3))
print('one chain  B=2: %.2f us/iteration' % (ctxs[0].time_sinkhorn(2, N, IT) * 2e3))
res = [None, None]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def work(i):
    with torch.cuda.stream(streams[i]):
        ctxs[i].time_sinkhorn(2, N, 50)
        res[i] = ctxs[i].time_sinkhorn(2, N, IT) * 2e3
for rep in range(2):
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    print('two chains B=2+2 on two streams: %.2f / %.2f us/iteration each (wall %.1f ms)' % (res[0], res[1], (time.time() - t0) * 1e3))

res = [None] * 4
streams = [torch.cuda.Stream() for _ in range(4)]
def work1(i):
    with torch.cuda.stream(streams[i]):
        ctxs[i].time_sinkhorn(1, N, 50)
        res[i] = ctxs[i].time_sinkhorn(1, N, IT) * 2e3
print('one chain  B=1: %.2f us/iteration' % (ctxs[0].time_sinkhorn(1, N, IT) * 2e3))
for rep in range(2):
    th = [threading.Thread(target=work1, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    print('four chains B=1 x4 on four streams: ' + ' / '.join('%.2f' % r for r in res) + ' us/iteration each')
