#!/bin/bash
# Sinkhorn iteration timing + OT parity tests (usage: gpurun -- bash tools/gpu_ot_ab.sh TAG)
TAG=${1:-ot}
mkdir -p gpurun_out
(timeout 240 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -rfE -p no:cacheprovider -k "score or matches" 2>&1 | tail -8) > gpurun_out/ot_$TAG.log 2>&1
timeout 120 python - >> gpurun_out/ot_$TAG.log 2>&1 <<PY
import sys, torch
sys.path.insert(0, 'tests')
from helpers import eval_config, make_hip_model
from imp_release_amd import synthetic
cfg = eval_config(n_layers=1)
m = make_hip_model('GM', cfg, synthetic.make_state_dict(cfg, model='GM', seed=1))
ctx = m._ensure_ctx()
for B, n in ((4, 2048), (1, 2048), (1, 1024), (1, 4096)):
    ms = min(ctx.time_sinkhorn(B, n, 100) for _ in range(3))
    print('B=%d n=%d  %.2f us / iteration' % (B, n, ms * 2e3))
PY
cat gpurun_out/ot_$TAG.log
