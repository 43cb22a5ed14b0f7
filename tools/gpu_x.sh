#!/bin/bash
# whole-sequence vs row-block kernels at the large-batch 64-row configs (cfg 2 / 3), after the row-block work of this round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 2 3; do
  for f in 0 1; do
    if [ $f = 1 ]; then export DTQN_FORCE_TILED=1; else unset DTQN_FORCE_TILED; fi
    timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/x_cfg${c}_$f.json 2> gpurun_out/x_cfg${c}_$f.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/x_cfg${c}_$f.json') if l.startswith('{')][0])
    print('cfg$c tiled=$f', round(d['value'],1), d['ms_per_step'])
except Exception as e:
    print('cfg$c tiled=$f failed', e, open('gpurun_out/x_cfg${c}_$f.err').read()[-400:])
PY
  done
done
