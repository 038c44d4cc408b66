#!/bin/bash
# the round's profile evidence in one call:  gpurun -- bash tools/gpu_profile_round.sh r5
#   1. rocprofv3 --kernel-trace --stats of bench.py (one step in flight, and the default calibrated run)
#   2. PMC passes, each in its own run (kernel-trace only beside --pmc): SQ busy / MFMA, FETCH_SIZE, WRITE_SIZE, GRBM
#   3. the kernel sequence of one step (tools/trace_sequence.py)
# results: gpurun_out/prof_<tag>_*  (copy what is to be judged into profiles/<round>/)
T=${1:-r5}; R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_one -o bench -- python $R/bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-batch1 --no-f32-mode --in-flight 1 2>/dev/null | tail -1) > $R/gpurun_out/prof_${T}_one_line.json 2>/dev/null
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_default -o bench -- python $R/bench.py --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1) > $R/gpurun_out/prof_${T}_default_line.json 2>/dev/null
cd $R
for k in one default; do
  S=$(find gpurun_out/prof_${T}_$k -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" gpurun_out/prof_${T}_${k}_kernel_stats.csv
done
F=$(find gpurun_out/prof_${T}_one -name "*kernel_trace.csv" | head -1)
python tools/trace_sequence.py "$F" > gpurun_out/prof_${T}_step_sequence.txt 2>&1
rm -rf gpurun_out/prof_${T}_one gpurun_out/prof_${T}_default
bash tools/gpu_pmc.sh $T "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" > /dev/null 2>&1
python tools/pmc_summary.py $T gpurun_out/prof_${T}_pmc_per_kernel.csv
python tools/traffic_json.py gpurun_out/prof_${T}_pmc_per_kernel.csv gpurun_out/prof_${T}_traffic.json > /dev/null
rm -rf gpurun_out/pmc_${T}_*/
head -8 gpurun_out/prof_${T}_one_kernel_stats.csv | cut -c1-150; head -6 gpurun_out/prof_${T}_pmc_per_kernel.csv | cut -c1-250
