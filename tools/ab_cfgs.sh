#!/bin/bash
# product build vs a variant at BASELINE configs 2-5 (and cfg 3 on the whole-sequence kernels): TD-updates/s
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in product "$@"; do
  if [ $v = product ]; then unset DTQN_HIP_LIB; else export DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_$v.so; fi
  for c in 2 3 3w 4 5; do
    cc=${c%w}; if [ $c = 3w ]; then export DTQN_TRAIN_TILED=0; else unset DTQN_TRAIN_TILED; fi
    timeout 200 python bench.py --config $cc --steps 300 --warmup 30 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/abc_${v}_$c.json 2> gpurun_out/abc_${v}_$c.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/abc_${v}_$c.json') if l.startswith('{')][0])
    print('$v cfg$c', round(d['value'],1), round(d['roofline']['whole_update_frac'],4), d['kernels_us'])
except Exception as e:
    print('$v cfg$c failed', e)
PY
  done
done
unset DTQN_HIP_LIB DTQN_TRAIN_TILED
