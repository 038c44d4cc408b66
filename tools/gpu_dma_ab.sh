#!/bin/bash
# one gpurun call: (1) tools/probe/attn_dma_check_<tag>.bin - the LDS-DMA staging variant of the attention kernel against the register-staged one, bit for bit, + timing of both;
# (2) the model-level GPU tests with IMP_ATTN_DMA=1; (3) bench.py A/B: default | IMP_ATTN_DMA=1 | IMP_ATTN_DMA=1 on the PP_DMA_SPREAD=0 build
R=$PWD; O=$R/gpurun_out; mkdir -p $O; L=$O/dma_ab_${1:-x}.log; : > $L
for tag in sp1 sp0 sp1_prio2 sp1_prio1 sp0_prio2; do
  echo "=== probe $tag" >> $L; timeout 120 tools/probe/attn_dma_check_$tag.bin >> $L 2>&1; echo "exit $?" >> $L
done
echo "=== GPU tests with IMP_ATTN_DMA=1" >> $L
(IMP_ATTN_DMA=1 timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ragged.py tests/test_gpu_ops.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -12) >> $L
echo "=== bench A/B" >> $L
for rep in 1 2; do
  for E in "-" "IMP_ATTN_DMA=1" "IMP_ATTN_DMA=1 IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_dma_sp0.so"; do
    [ "$E" = "-" ] && E=""
    (env $E timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-batch1 --no-f32-mode 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-100s pairs/s %.1f  ms/step %.3f  one-in-flight %.1f  attn %.2f us  frac %.3f  sclk %.0f' % ('$E' or '(default)', d['value'], d['ms_per_step'], d['one_step_in_flight']['value'], r['launch_ms']*1e3, r['frac'], r.get('sclk_mhz_observed') or 0))
") >> $L 2>&1
  done
done
cat $L
