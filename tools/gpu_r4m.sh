#!/bin/bash
# native EIMP lock-step driver: tests, rates, divergence statistics
TAG=${1:-r4m}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=400 -x -k "eimp" 2>&1 | tail -25) > $O/${TAG}_tests.log 2>&1
for cfg in "--workers 3" "--lockstep 4 --workers 1" "--lockstep 4 --workers 2" "--lockstep 4 --workers 3" "--lockstep 2 --workers 3"; do
  echo "== EIMP $cfg" >> $O/${TAG}_loops.log
  (timeout 400 python tools/eval_synthetic.py --hard --model EIMP --pairs 192 $cfg 2>&1 | tail -1 | cut -c1-420) >> $O/${TAG}_loops.log
done
cat $O/${TAG}_tests.log $O/${TAG}_loops.log
