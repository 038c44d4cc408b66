#!/usr/bin/env python3
"""How does parity degrade as the network gets worse conditioned - and how much of that is the REFERENCE's own fp32 noise?

Build-container tool (imports /root/reference; nothing at test / bench time does).  VERDICT r4 #5a: all committed parity evidence sat on
weights chosen to be well conditioned; with sharper attention (larger q / k projections) every rounding of the logits is amplified by the
softmax, by 18 layers and by the exp of the Sinkhorn, and two fp32 evaluations of the REFERENCE ITSELF stop agreeing.  This script sweeps the
q / k gain of the trained-style synthetic weights (synthetic.make_state_dict(style='trained', qk_gain=g)) at N = 1024 (GM, 9 iterations,
100 Sinkhorn iterations, only_last) and records, per gain,

  * the fp64 run of the reference (model.double(), float64 inputs): indices0 / mscores0 - the yardstick,
  * the reference's own fp32 noise against it: fp32 with 1 thread and with 8 threads (different summation orders inside MKL / ATen),
    each as (index disagreements, max |mscore - fp64| over agreeing keypoints), and 1 thread vs 8 threads against each other,

into tests/golden/conditioning_n1024.npz.  tests/test_gpu_parity.py::test_parity_under_conditioning then asserts, on the GPU and in both
precisions, that the HIP path disagrees with the fp64 yardstick NO MORE than the reference's own fp32 evaluations do (index disagreements
<= the larger of the reference's two counts + the number of keypoints the reference itself leaves unstable, score deviation <=
max(1e-4, 2 x the larger of the reference's two)) - which replaces the prose
justification of the `low_score_flips` tolerance with a measurement of where the reference stops defining the answer.

    python tools/parity_vs_conditioning.py [N]        # writes tests/golden/conditioning_n<N>.npz (default 1024; 2048: gains 1, 2, 3), prints the table
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
_ones = torch.ones


def _ones_cpu(*a, **k):                       # nets/layers.py:41-44 hard-codes device='cuda'
    if str(k.get('device', '')) == 'cuda':
        k['device'] = 'cpu'
    return _ones(*a, **k)


torch.ones = _ones_cpu
cv2 = types.ModuleType('cv2'); cv2.USAC_MAGSAC = 38; cv2.RANSAC = 8
sys.modules['cv2'] = cv2
from nets.gm import GM                        # noqa: E402  (the reference)
from imp_release_amd import synthetic         # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024          # round 6: `python tools/parity_vs_conditioning.py 2048` = the sweep at the headline size (gains 1 ... 3)
GAINS = [1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 5.0] if N <= 1024 else [1.0, 2.0, 3.0]
WSEED, DSEED = 21, 511
CFG = {'descriptor_dim': 256, 'sinkhorn_iterations': 100, 'match_threshold': 0.2, 'with_sinkhorn': True, 'n_layers': 9,
       'GNN_layers': ['self', 'cross'] * 9, 'ac_fn': 'relu', 'norm_fn': 'in', 'n_min_tokens': 256}


def run(sd, data, dtype, threads):
    torch.set_num_threads(threads)
    m = GM(CFG).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dtype)
    d = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in data.items()}
    with torch.no_grad():
        out = m.produce_matches(d, p=0.2, only_last=True)
    return out['indices0'][-1][0].numpy().astype(np.int64), out['mscores0'][-1][0].double().numpy()


def versus(a, b):
    bad = int((a[0] != b[0]).sum())
    agree = (a[0] == b[0])
    return bad, float(np.abs(a[1] - b[1])[agree].max(initial=0.0))


def main():
    pair = synthetic.make_correlated_pair(N, N, seed=DSEED)
    data = {k: torch.from_numpy(v) for k, v in pair.items() if k != 'image_shape'}
    data['image0'] = data['image1'] = torch.zeros(pair['image_shape'])
    arrays = {}
    print(f'{"qk gain":>8s} {"matched64":>9s} | {"ref fp32/1t vs fp64":>22s} | {"ref fp32/8t vs fp64":>22s} | {"ref 1t vs 8t":>18s}')
    for g in GAINS:
        sd = synthetic.make_state_dict(CFG, 'GM', seed=WSEED, style='trained', qk_gain=g)
        r64 = run(sd, data, torch.float64, 8)
        r1 = run(sd, data, torch.float32, 1)
        r8 = run(sd, data, torch.float32, 8)
        v1, v8, v18 = versus(r1, r64), versus(r8, r64), versus(r1, r8)
        tag = f'g{int(round(g * 10)):02d}'
        arrays[f'{tag}_indices0'] = r64[0]; arrays[f'{tag}_mscores0'] = r64[1]
        # keypoints the reference itself does not pin down: an fp32 evaluation of it moves their mscore by more than 1e-3 (ten times the parity bar)
        unstable = int(((np.abs(r1[1] - r64[1]) > 1e-3) | (np.abs(r8[1] - r64[1]) > 1e-3)).sum())
        arrays[f'{tag}_ref_noise'] = np.array([v1[0], v1[1], v8[0], v8[1], v18[0], v18[1], unstable], dtype=np.float64)
        print(f'{g:8.1f} {int((r64[0] >= 0).sum()):9d} | {v1[0]:6d} idx {v1[1]:10.2e} | {v8[0]:6d} idx {v8[1]:10.2e} | {v18[0]:4d} idx {v18[1]:9.2e} | unstable keypoints {unstable}', flush=True)
    spec = {'model': 'GM', 'config': {'n_layers': 9, 'sinkhorn_iterations': 100}, 'n': N, 'wseed': WSEED, 'dseed': DSEED, 'style': 'trained',
            'gains': GAINS, 'call': {'p': 0.2, 'only_last': True},
            'ref_noise_columns': ['idx fp32/1t vs fp64', 'dms fp32/1t vs fp64', 'idx fp32/8t vs fp64', 'dms fp32/8t vs fp64', 'idx 1t vs 8t', 'dms 1t vs 8t', 'keypoints whose mscore an fp32 evaluation of the reference moves by > 1e-3']}
    arrays['spec_json'] = np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8)
    out = os.path.join(ROOT, 'tests', 'golden', f'conditioning_n{N}.npz')
    np.savez_compressed(out, **arrays)
    print(f'wrote {out} ({os.path.getsize(out) / 1024:.1f} KiB)')


if __name__ == '__main__':
    main()
