"""does the ping-pong attention kernel get faster when K / V arrive pre-split (hi | lo f16 halves written by the producer) and are
staged by plain copy instead of being converted by every workgroup?  Run once with IMP_ATTN_KV_PLANES=0 and once with =1 (the switch is read once
per process): prints the launch time at B = 4, N = 2048 over ~4 s of back-to-back launches with the clock / power samples taken meanwhile, and
writes the attention output of a fixed input so that the two runs can be compared bit for bit (tools/gpu_attn_kv.sh)."""
import os, subprocess, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
mode = os.environ.get('IMP_ATTN_KV_PLANES', '0')
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
pair = synthetic.make_correlated_pair(2048, 2048, seed=1, batch=4)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
m.produce_matches(d, p=0.2, only_last=True)
g = torch.Generator().manual_seed(3)
qkv = torch.randn(2, 1500, 768, generator=g).cuda()
out, lse = ctx.op_attention(qkv, qkv)
np.save(os.path.join(ROOT, 'gpurun_out', f'attn_kvp{mode}_out.npy'), out.cpu().numpy())

def sample(stop, rows):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5)
            rows.append(r.stdout.strip().splitlines()[-1][:300])
        except Exception as e:
            rows.append(f'rocm-smi failed: {e}')
        time.sleep(0.3)

if '--quick' in sys.argv:          # under rocprofv3 --pmc: a handful of launches only
    print(f'kv_planes={mode}: {ctx.time_attention(4, 2048, 10) * 1e3:.2f} us')
    sys.exit(0)
stop, rows = threading.Event(), []
th = threading.Thread(target=sample, args=(stop, rows)); th.start()
t0 = time.time(); ms = []
while time.time() - t0 < 4.0:
    ms.append(ctx.time_attention(4, 2048, 200))
stop.set(); th.join()
ms = np.array(ms) * 1e3
print(f'kv_planes={mode}: launch min {ms.min():.2f} median {np.median(ms):.2f} max {ms.max():.2f} us over {len(ms)} x 200 launches')
for r in rows[2:8]:
    print('   ', r)
