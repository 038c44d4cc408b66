for v in "" pkadd cneg; do
  if [ -n "$v" ]; then export IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_$v.so; else unset IMP_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['one_step_in_flight']['value'], d['roofline']['launch_ms'])"
done
