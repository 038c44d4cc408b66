#!/bin/bash
# PMC passes over the SuperPoint front-end (own runs: --pmc with kernel-trace only).  usage: gpurun -- bash tools/gpu_sp_pmc.sh
R=$PWD
mkdir -p gpurun_out/sp
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  (timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sppmc_$i -o sp -- python $R/tools/probe/sp_time.py 480 640 1024 3 2>&1 | tail -1) > /tmp/sppmc_$i.log 2>&1
  f=$(find /tmp/sppmc_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print('no counter file', e); sys.exit(0)
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if 'sp_' not in name:
        continue
    key = (name, r.get('Grid_Size', '?'))
    d = agg.setdefault(key, collections.OrderedDict())
    d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for (name, grid), d in agg.items():
    print(f'{name:36s} grid {grid:>8s} ' + '  '.join(f'{k}={sorted(v)[len(v) // 2]:.4g}' for k, v in d.items()))
PY
done 2>&1 | tee $R/gpurun_out/sp/sp_480x640_pmc.txt
