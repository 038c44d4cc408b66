#!/bin/bash
TAG=${1:-r4n}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for cfg in "--lockstep 2 --workers 2" "--lockstep 2 --workers 3" "--lockstep 2 --workers 4" "--lockstep 3 --workers 2" "--lockstep 3 --workers 3" "--lockstep 1 --workers 4" "--lockstep 2 --workers 6" "--lockstep 4 --workers 1"; do
  echo "== EIMP $cfg" >> $O/${TAG}_loops.log
  (timeout 400 python tools/eval_synthetic.py --hard --model EIMP --pairs 192 $cfg 2>&1 | tail -1 | cut -c1-260 | sed 's/"includes.*//') >> $O/${TAG}_loops.log
done
for cfg in "--lockstep 4 --workers 3" "--lockstep 2 --workers 4" "--lockstep 4 --workers 2"; do
  echo "== IMP $cfg" >> $O/${TAG}_loops.log
  (timeout 400 python tools/eval_synthetic.py --hard --model IMP --pairs 192 $cfg 2>&1 | tail -1 | cut -c1-260 | sed 's/"includes.*//') >> $O/${TAG}_loops.log
done
cat $O/${TAG}_loops.log
