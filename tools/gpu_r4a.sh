#!/bin/bash
# round 4, first look at the fused layer kernel:  gpurun -- bash tools/gpu_r4a.sh TAG
#   kernel-level tests, model-level bit-identity / time-out tests, timing probe, same-box bench A/B (fused on / off, lane on / off)
TAG=${1:-r4a}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -rfE -p no:cacheprovider -x --timeout=120 -k "fused" 2>&1 | tail -15) > $O/${TAG}_ops.log 2>&1
(timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=200 -k "fused or in_flight or chained" 2>&1 | tail -25) > $O/${TAG}_parity.log 2>&1
(timeout 200 python - <<'PY'
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1); sd = synthetic.make_state_dict(cfg, 'GM', seed=0)
m = make_hip_model('GM', cfg, sd); ctx = m._ensure_ctx()
for B, N in ((4, 2048), (3, 2048), (4, 1500)):
    pair = synthetic.make_correlated_pair(N, N, seed=1, batch=B)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k != 'image_shape'}
    d['image0'] = d['image1'] = torch.zeros(pair['image_shape'], device='cuda')
    m.produce_matches(d, p=0.2, only_last=True)
    for rep in range(2):
        t = {w: ctx.time_layer_gemm(B, N, w, -2) * 1e3 for w in (0, 1, 2, 3, 4)}
        print(f'B={B} N={N}: QKV {t[0]:.1f}  MLP0 {t[1]:.1f}  MLP3 {t[2]:.1f}  MLP3+QKV chained {t[3]:.1f}  | two launches {t[1] + t[3]:.1f} us  FUSED {t[4]:.1f} us', flush=True)
PY
) > $O/${TAG}_time.log 2>&1
REPS=1 STEPS=40 bash tools/gpu_ab.sh $TAG "-" "IMP_WF_FUSED=0" "IMP_OT_LANE=1" "IMP_WF_FUSED=0 IMP_OT_LANE=1" > /dev/null 2>&1
cat $O/${TAG}_ops.log $O/${TAG}_parity.log $O/${TAG}_time.log $O/ab_$TAG.log
