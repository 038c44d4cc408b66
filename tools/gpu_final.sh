#!/bin/bash
# driver-like final check + the evidence files of a round:  tools/gpu_final.sh <round-tag>   (one gpurun call)
#   GPU tests, smoke, default bench (with CPU baseline), rocprofv3 kernel stats of the bench workload (one step in flight and
#   the default), of BASELINE configs[1] (N=1024, batch 1) and of the EIMP loop; everything lands in gpurun_out/final_<tag>_*
R=$PWD; T=${1:-x}; O=$R/gpurun_out; mkdir -p $O
(timeout 500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -6) > $O/final_${T}_pytest.log 2>&1
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/final_${T}_smoke.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -1) > $O/final_${T}_bench.json 2>&1
cd /tmp && export TMPDIR=/tmp
# per-kernel evidence is taken with ONE step in flight: overlapped kernels share the GPU, their durations are then not those of a full-chip launch
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_${T}_prof1 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --in-flight 1 2>&1 | tail -1) > $O/final_${T}_rocprof1.log 2>&1
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_${T}_prof2 -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/final_${T}_rocprof2.log 2>&1
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_${T}_c2 -o c2 -- python $R/tools/probe/c2_run.py 1024 1 2>&1 | tail -1) > $O/final_${T}_c2.log 2>&1
cd $R
(timeout 200 python tools/gpu_configs.py 2>&1 | tail -7) > $O/final_${T}_configs.log 2>&1
(timeout 300 python tools/eval_synthetic.py --pairs 2000 --model IMP --hard --workers 3 --lockstep 4 --pose gpu 2>&1 | tail -1; timeout 300 python tools/eval_synthetic.py --pairs 1000 --model IMP --hard --workers 3 --pose gpu 2>&1 | tail -1; timeout 300 python tools/eval_synthetic.py --pairs 2000 --model EIMP --hard --workers 3 --pose gpu 2>&1 | tail -1; for M in IMP EIMP; do timeout 300 python tools/eval_synthetic.py --pairs 1000 --model $M --kpts 2048 --workers 3 --pose gpu 2>&1 | tail -1; done; timeout 200 python tools/eval_synthetic.py --pairs 200 --model EIMP --kpts 2048 --workers 3 --pose none --weights uniform 2>&1 | tail -1) > $O/final_${T}_loop.log 2>&1
# round 4: the fused layer launch - phase cycles (-DWF_PROFILE variant, if built), same-box A/B against the two-launch layers, lock-step rates
([ -f imp-release_amd/csrc/variants/libimp_hip_wfprof.so ] && IMP_WF_PROF=1 IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_wfprof.so timeout 200 python tools/probe/fused_time.py 2>&1 | grep -v amdgpu.ids | tail -9; timeout 200 python tools/probe/fused_time.py 2>&1 | grep -v amdgpu.ids | tail -3) > $O/final_${T}_fused.log 2>&1
REPS=2 STEPS=60 bash tools/gpu_ab.sh final_$T "-" "IMP_WF_FUSED=0" "IMP_WF_FUSED=0 IMP_OT_LANE=1" > /dev/null 2>&1
(timeout 300 python tools/probe/lockstep_profile.py 2>&1 | grep -v amdgpu.ids | grep "lockstep [0-9]") > $O/final_${T}_lockstep.log 2>&1
# pose step: per-kernel times of a call (n = 200 / 1000 / 3000), phase stamps of the five-point solver (-DFP_PROFILE variant, if built), and the
# same-box A/B of the iterative loops against the library with the first round-3 pose step (variants/libimp_hip_oldpose.so, if built)
(python tools/probe/pose_time.py 2>&1 | grep "n=" | cut -c1-130; cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_${T}_pose -o p -- python $R/tools/probe/pose_time.py > /dev/null 2>&1; cd $R; python - <<EOF
import csv
for r in list(csv.DictReader(open('$O/final_${T}_pose/p_kernel_stats.csv')))[:8]:
    print('%-64s calls %4s  avg %9.1f us  %5s %%' % (r['Name'].replace('(anonymous namespace)::','')[:64], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
EOF
[ -f imp-release_amd/csrc/variants/libimp_hip_fpprof.so ] && IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_fpprof.so python tools/probe/pose_phases.py 2>&1 | grep fivept | head -4) > $O/final_${T}_pose.log 2>&1
([ -f imp-release_amd/csrc/variants/libimp_hip_oldpose.so ] && for r in 1 2; do for L in new old; do if [ $L = old ]; then export IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_oldpose.so; else unset IMP_HIP_LIB; fi; for M in IMP EIMP; do echo -n "$L pose library, $M loop, 600 pairs, 3 in flight: "; timeout 300 python tools/eval_synthetic.py --pairs 600 --model $M --kpts 2048 --workers 3 --pose gpu 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['pairs_per_s'],1), 'pairs/s  AUC@5/10', d['report']['auc@5'], d['report']['auc@10'])"; done; done; done; unset IMP_HIP_LIB) > $O/final_${T}_pose_ab.log 2>&1
# PMC passes (own runs, --pmc with --kernel-trace only): matrix-pipe / VALU / LDS counters and the HBM traffic of every product kernel
bash tools/gpu_pmc.sh $T "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" > $O/final_${T}_pmc.log 2>&1
python tools/pmc_summary.py $T $O/final_${T}_pmc_per_kernel.csv
(for K in 1 2 3; do timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-batch1 --in-flight $K 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in-flight $K: %.1f pairs/s' % d['value'])"; done) > $O/final_${T}_inflight.log 2>&1
cat $O/final_${T}_fused.log $O/ab_final_$T.log $O/final_${T}_lockstep.log; tail -4 $O/final_${T}_pytest.log; tail -2 $O/final_${T}_smoke.log; cut -c1-1800 $O/final_${T}_bench.json; echo; head -9 $O/final_${T}_prof1/bench_kernel_stats.csv | cut -c1-150; cat $O/final_${T}_configs.log; cat $O/final_${T}_loop.log $O/final_${T}_inflight.log; cat $O/final_${T}_pose.log $O/final_${T}_pose_ab.log; head -12 $O/final_${T}_pmc_per_kernel.csv | cut -c1-250
