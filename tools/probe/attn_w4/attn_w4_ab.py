"""same-process, interleaved A/B of the two f16x3 attention kernels at the bench geometry (HIP events on the launch stream, `reps` back-to-back launches per sample,
shader clock from the in-kernel probe): 8-wave ping-pong (attention_f16x3.hip) vs one wave per SIMD (attention_f16x3_w4.hip).
    python tools/probe/attn_w4_ab.py [B N reps rounds]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import eval_config, make_hip_model
from imp_release_amd import _lib, synthetic
B, N, reps, rounds = (int(v) for v in (sys.argv[1:5] + ['4', '2048', '20', '8'][len(sys.argv) - 1:]))
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
m.produce_matches(d, p=0.2, only_last=True)
word = ctypes.c_int.in_dll(_lib.lib(), 'imp_attn_w4_override')
for _ in range(3):
    ctx.time_attention(B, N, reps)
res = {0: [], 1: []}
for r in range(rounds):
    for v in (0, 1):
        word.value = v
        ms, mhz = ctx.time_attention_clock(B, N, reps)
        res[v].append((ms * 1e3, mhz))
word.value = -1
for v, name in ((0, 'ping-pong 8 waves'), (1, 'one wave per SIMD')):
    us = sorted(x[0] for x in res[v]); mh = sorted(x[1] for x in res[v])
    print(f'{name:20s} B={B} N={N}: median {us[len(us) // 2]:.2f} us (min {us[0]:.2f}, max {us[-1]:.2f}), clock median {mh[len(mh) // 2]:.0f} MHz')
a, b = sorted(x[0] for x in res[0]), sorted(x[0] for x in res[1])
print(f'one wave per SIMD / ping-pong = {b[len(b) // 2] / a[len(a) // 2]:.4f}')
