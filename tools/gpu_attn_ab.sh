#!/bin/bash
# A/B of the attention kernels (usage: gpurun -- bash tools/gpu_attn_ab.sh TAG): parity of the large-grid path, then timing
TAG=${1:-ab}
mkdir -p gpurun_out
(timeout 240 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -rfE -p no:cacheprovider -k "attention" 2>&1 | tail -15) > gpurun_out/attn_$TAG.log 2>&1
for v in 0 1; do
  IMP_ATTN_VARIANT=$v timeout 120 python - >> gpurun_out/attn_$TAG.log 2>&1 <<PY
import sys, torch
sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1)
m = make_hip_model('GM', cfg, synthetic.make_state_dict(cfg, model='GM', seed=1))
ctx = m._ensure_ctx()
for B, n in ((4, 2048), (1, 4096), (8, 2048)):
    try:
        ms = min(ctx.time_attention(B, n, 20) for _ in range(3))
        print('variant $v  B=%d n=%d  %.1f us  %.1f TF' % (B, n, ms * 1e3, 2 * 4 * B * 2 * n * n * 64 * 2 / ms / 1e9))
    except Exception as e:
        print('variant $v B=%d n=%d failed: %s' % (B, n, e))
PY
done
cat gpurun_out/attn_$TAG.log
