#!/bin/bash
TAG=${1:-r4q}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=400 -x -k "lockstep" 2>&1 | tail -5) > $O/${TAG}_tests.log 2>&1
echo "== IMP_LOOP_IMMEDIATE=0 (deferred exit decisions, round 4a)" >> $O/${TAG}_sweep.log
(IMP_LOOP_IMMEDIATE=0 timeout 600 python tools/probe/c5_sweep.py 1200 4x3,3x3 2>&1 | grep "lockstep" | grep DGNNS) >> $O/${TAG}_sweep.log
echo "== default (immediate)" >> $O/${TAG}_sweep.log
(timeout 600 python tools/probe/c5_sweep.py 1200 4x3,3x3,4x4 2>&1 | grep "lockstep") >> $O/${TAG}_sweep.log
echo "== default + group_similar" >> $O/${TAG}_sweep.log
(timeout 600 python tools/probe/c5_sweep.py 1200 4x3,3x3,4x4 similar 2>&1 | grep "lockstep") >> $O/${TAG}_sweep.log
cat $O/${TAG}_tests.log $O/${TAG}_sweep.log
