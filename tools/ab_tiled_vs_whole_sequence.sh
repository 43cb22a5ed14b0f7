#!/bin/bash
# whole-sequence vs row-block kernels at cfg 3 (D = 128, 64-row contexts, B = 512), after the row-block work of this round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 3; do
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
export DTQN_FORCE_TILED=1
export TMPDIR=/tmp
rm -rf /tmp/kt3
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -- python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-env-rate --no-other-configs > /dev/null 2>&1
DB=$(find /tmp/kt3 -name '*results.db' | head -1)
python tools/rocpd_summary.py "$DB" gpurun_out/x_kernel_stats_cfg3_tiled.md > /dev/null
head -22 gpurun_out/x_kernel_stats_cfg3_tiled.md
