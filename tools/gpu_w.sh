#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 2 3 4 5; do
    timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/w_cfg${c}.json 2> gpurun_out/w_cfg${c}.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/w_cfg${c}.json') if l.startswith('{')][0])
    print('cfg$c', round(d['value'],1), d['ms_per_step'])
except Exception as e:
    print('cfg$c failed', e)
PY
done
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_bag.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3
