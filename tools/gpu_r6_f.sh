#!/bin/bash
R=$PWD; O=$R/gpurun_out/r6f; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_hard_loops.py tests/test_gpu_batch_invariance.py tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-400) > $O/tests.log 2>&1
grep -n "^E  .*Error\|^E   *\(eimp\|imp\) pair\|passed\|failed\|^FAILED" $O/tests.log | cut -c1-400; grep -A40 "harder two-view" $O/tests.log | cut -c1-300
