#!/bin/bash
# A/B on one box: attention kernel with the loop over a unit's shares (library) against the kernel before it (variants/libimp_hip_preserial.so); then the c5 loops
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O; : > $O/ab.log
for i in 1 2 3; do
  (echo -n "new  "; timeout 120 python tools/probe/attn_time.py 2>&1 | grep "^attention") >> $O/ab.log
  (echo -n "pre  "; IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_preserial.so timeout 120 python tools/probe/attn_time.py 2>&1 | grep "^attention") >> $O/ab.log
done
for L in imp eimp; do for W in 1 4; do (timeout 300 python tools/probe/c5_probe.py $L $W 600 4 2>&1 | grep "^c5") >> $O/ab.log; done; done
cat $O/ab.log
