#!/bin/bash
# linear and dY W kernels: 64- vs 32-row workgroups at cfg 4 / 5 (DTQN_GEMM_ROWS), default policy last, then parity of the tiled path
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 4 5; do
  for r in 64 32 auto; do
    if [ $r = auto ]; then unset DTQN_GEMM_ROWS; else export DTQN_GEMM_ROWS=$r; fi
    timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/y_cfg${c}_$r.json 2> gpurun_out/y_cfg${c}_$r.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/y_cfg${c}_$r.json') if l.startswith('{')][0])
    print('cfg$c rows=$r', round(d['value'],1), d['ms_per_step'])
except Exception as e:
    print('cfg$c rows=$r failed', e)
PY
  done
done
unset DTQN_GEMM_ROWS
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_bag.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -3
