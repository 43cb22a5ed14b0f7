#!/bin/bash
# split weight-gradient kernel: (split, tile) pairs in contiguous runs per XCD (DTQN_WGRAD_XCD=1, default) vs pair = block (0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 2 3 4 5; do
  for x in 0 1; do
    export DTQN_WGRAD_XCD=$x
    timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/v_cfg${c}_$x.json 2> gpurun_out/v_cfg${c}_$x.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/v_cfg${c}_$x.json') if l.startswith('{')][0])
    print('cfg$c xcd=$x', round(d['value'],1), d['ms_per_step'])
except Exception as e:
    print('cfg$c xcd=$x failed', e)
PY
  done
done
unset DTQN_WGRAD_XCD
timeout 600 python -m pytest tests/test_gpu_td.py -x -q -m gpu 2>&1 | tail -3
