#!/bin/bash
# pre-split K/V experiment of the attention kernel (tools/probe/attn_kv_planes.py): timing + clock / power, bit comparison, VALU / MFMA counters
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for M in 0 1 0 1; do IMP_ATTN_KV_PLANES=$M timeout 120 python tools/probe/attn_kv_planes.py 2>&1 | grep -v amdgpu.ids; done > $O/attn_kvp_time.log 2>&1
python - <<PY >> $O/attn_kvp_time.log 2>&1
import numpy as np
a, b = np.load('$O/attn_kvp0_out.npy'), np.load('$O/attn_kvp1_out.npy')
print('outputs bit-identical:', bool(np.array_equal(a, b)), ' max|diff|', float(np.abs(a - b).max()))
PY
cd /tmp && export TMPDIR=/tmp
for M in 0 1; do
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (IMP_ATTN_KV_PLANES=$M timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_kvp${M}_$i -o bench -- python $R/tools/probe/attn_kv_planes.py --quick 2>&1 | tail -1) >> $O/attn_kvp_pmc.log 2>&1
  done
done
cd $R
for M in 0 1; do echo "== kv_planes=$M" >> $O/attn_kvp_pmc.log; python tools/pmc_summary.py kvp$M 2>/dev/null | grep -i "attn_f16x3_pp\|^kernel" | head -4 >> $O/attn_kvp_pmc.log; done
cat $O/attn_kvp_time.log; tail -20 $O/attn_kvp_pmc.log
