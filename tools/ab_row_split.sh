#!/bin/bash
# row-split policy at large batches: cfg 2 / 3 under DTQN_ROW_SPLIT = default / 1 (two slices) / 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 2 3; do
  for rs in default 1 4; do
    if [ $rs = default ]; then unset DTQN_ROW_SPLIT; else export DTQN_ROW_SPLIT=$rs; fi
    timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/u_cfg${c}_$rs.json 2> gpurun_out/u_cfg${c}_$rs.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/u_cfg${c}_$rs.json') if l.startswith('{')][0])
    print('cfg$c rs=$rs', round(d['value'],1), d['ms_per_step'], d['roofline'].get('kernels_us'))
except Exception as e:
    print('cfg$c rs=$rs failed', e)
PY
  done
done
