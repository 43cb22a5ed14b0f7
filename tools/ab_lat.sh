#!/bin/bash
# A/B of engine builds at cfg 1: the product build vs variants built by tools/build_variant.py (DTQN_HIP_LIB)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in product "$@"; do
  if [ $v = product ]; then unset DTQN_HIP_LIB; else export DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_$v.so; fi
  timeout 200 python bench.py --steps 3000 --warmup 300 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab_$v.json') if l.startswith('{')][0])
    print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'], d['update_us_median'])
except Exception as e:
    print('$v failed', e, open('gpurun_out/ab_$v.err').read()[-500:])
PY
done
unset DTQN_HIP_LIB
