#!/bin/bash
# round 6, first GPU call: the new invariance / recovery / hard-set tests first (fast feedback), then the whole suite, the lock-step diff on 96 hard pairs,
# the default bench line
R=$PWD; O=$R/gpurun_out/r6a; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_batch_invariance.py -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -60) > $O/invariance.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_hard_loops.py -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -60) > $O/hard_loops.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider --deselect tests/test_gpu_batch_invariance.py --deselect tests/test_gpu_hard_loops.py 2>&1 | tail -80) > $O/suite.log 2>&1
(timeout 600 python tools/probe/eimp_lockstep_diff.py 96 2>&1 | tail -40) > $O/lockstep_diff.log 2>&1
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench.json 2>&1
echo ---- invariance; tail -30 $O/invariance.log | cut -c1-400
echo ---- hard loops; tail -30 $O/hard_loops.log | cut -c1-400
echo ---- suite; tail -40 $O/suite.log | cut -c1-400
echo ---- lockstep diff; tail -12 $O/lockstep_diff.log | cut -c1-300
echo ---- bench; cut -c1-2500 $O/bench.json
