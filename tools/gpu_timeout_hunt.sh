#!/bin/bash
# after the fix of the half-sum exchange: Sinkhorn tests, timing, and the hunt that met 3 voided launches in 12 runs before the fix
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O; : > $O/hunt.log
(timeout 900 python -m pytest tests/test_gpu_resident_ot.py tests/test_gpu_batch_invariance.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -2) > $O/tests.log 2>&1
(IMP_OPTIONS=probe_prof=1 timeout 200 python tools/probe/sk_prof.py 4 2048 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo) >> $O/tests.log
cat $O/tests.log | cut -c1-400
for i in 1 2 3 4 5 6 7 8; do
  for L in eimp imp; do (timeout 400 python tools/probe/c5_probe.py $L 4 4000 4 2>&1 | grep "^c5\|^replica") >> $O/hunt.log; done
done
cat $O/hunt.log | cut -c1-600
