# Sinkhorn iteration time per shape for the default library and for variant builds (tools/build_variant.sh)
for v in "" "$@"; do
  if [ -n "$v" ]; then export IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_$v.so; else unset IMP_HIP_LIB; fi
  echo "== ${v:-default}"; timeout 120 python tools/probe/sk_time.py 2>&1 | tail -6 | head -4
done
