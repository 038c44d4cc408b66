#!/bin/bash
TAG=${1:-r4o}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for cfg in "--model EIMP --lockstep 4 --workers 3" "--model EIMP --lockstep 2 --workers 3" "--model IMP --lockstep 4 --workers 3" "--model IMP --lockstep 2 --workers 4" "--model IMP --lockstep 2 --workers 3"; do
  echo "== $cfg" >> $O/${TAG}_loops.log
  (timeout 400 python tools/eval_synthetic.py --hard --pairs 192 $cfg 2>&1 | tail -1 | cut -c1-300 | sed 's/"includes.*//') >> $O/${TAG}_loops.log
  echo "== $cfg IMP_OT_LOCAL=0" >> $O/${TAG}_loops.log
  (IMP_OT_LOCAL=0 timeout 400 python tools/eval_synthetic.py --hard --pairs 192 $cfg 2>&1 | tail -1 | cut -c1-300 | sed 's/"includes.*//') >> $O/${TAG}_loops.log
done
cat $O/${TAG}_loops.log
