#!/bin/bash
# round 5, GPU session 6: BASELINE config 2 (batch 256) under the library's A/B knobs: waves per workgroup, weights through LDS or not, the row-block family
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/s6; mkdir -p $OUT
B="--no-other-configs --no-cpu-baseline --no-env-rate"
rate() {   # tag
  timeout 120 python bench.py --config 2 --steps 300 --warmup 40 $B 2>$OUT/rate_$1.err > $OUT/rate_$1.json
  python -c "
import json,sys
try:
    d=json.loads([l for l in open('$OUT/rate_$1.json') if l.startswith('{')][-1]); print('rate $1', round(d['value'],1), 'upd/s', round(d['ms_per_step']*1e3,2), 'us', d.get('kernels_us'))
except Exception as e: print('rate $1 failed', e, open('$OUT/rate_$1.err').read()[-400:])"
}
rate default
DTQN_WAVES=4 rate waves4
DTQN_WAVES=4 DTQN_WL=0 rate waves4_nowl
DTQN_WL=0 rate nowl
DTQN_WAVES=16 rate waves16
DTQN_TRAIN_TILED=1 rate tiled
DTQN_ROW_SPLIT=2 rate rowsplit2
rate default2
