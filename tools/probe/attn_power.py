"""Clock / power trace while ONE kernel runs back to back (back or retire the "power-limited" reading of the
attention kernel): rocm-smi is sampled every ~0.25 s during ~4 s of (a) the attention launch of the benchmark (B=4, N=2048),
(b) the QKV GEMM of the same size, (c) the attention launch on a quarter of the chip (B=1)."""
import subprocess, sys, threading, time
import torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
for B in (4, 1):
    pair = synthetic.make_correlated_pair(2048, 2048, seed=1, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    m.produce_matches(d, p=0.2, only_last=True)

def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5)
            out.append((time.time(), r.stdout.strip()))
        except Exception as e:
            out.append((time.time(), f'rocm-smi failed: {e}'))
        time.sleep(0.25)

def run(tag, fn):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); ms = []
    while time.time() - t0 < 4.0:
        ms.append(fn())
    stop.set(); th.join()
    print(f'== {tag}: launch {min(ms) * 1e3:.1f} .. {max(ms) * 1e3:.1f} us over {len(ms)} timings')
    for t, txt in out:
        lines = txt.splitlines()
        print(f'  t={t - t0:5.2f}s  ' + (' | '.join(lines[:1] + lines[1:2]) if len(lines) > 1 else txt)[:400])

run('attention B=4 N=2048 (256 workgroups = every CU)', lambda: ctx.time_attention(4, 2048, 200))
run('QKV GEMM B=4 N=2048', lambda: ctx.time_layer_gemm(4, 2048, 0, -1, 200))
run('attention B=1 N=2048 key-split off would be 64 workgroups; as dispatched', lambda: ctx.time_attention(1, 2048, 200))
run('idle', lambda: (time.sleep(0.2), 0.0)[1])
