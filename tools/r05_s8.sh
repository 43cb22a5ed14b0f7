#!/bin/bash
# round 5, GPU session 8: BASELINE config 2 (d_model 64, batch 256) trained on the row-block family (variant t64: dtqn_td_prefers_tiled admits d_model 64 under DTQN_TRAIN_TILED=1)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/s8; mkdir -p $OUT
B="--no-other-configs --no-cpu-baseline --no-env-rate"
V=$GRAFT_REPO_ROOT/tools/variants
rate() {   # tag lib config
  DTQN_HIP_LIB=$2 timeout 120 python bench.py --config $3 --steps 300 --warmup 40 $B 2>$OUT/rate_$1.err > $OUT/rate_$1.json
  python -c "
import json,sys
try:
    d=json.loads([l for l in open('$OUT/rate_$1.json') if l.startswith('{')][-1]); print('rate $1', round(d['value'],1), 'upd/s', round(d['ms_per_step']*1e3,2), 'us', d.get('kernels_us'))
except Exception as e: print('rate $1 failed', e, open('$OUT/rate_$1.err').read()[-600:])"
}
rate prod $GRAFT_REPO_ROOT/dtqn_amd/csrc/libdtqn_hip.so 2
DTQN_TRAIN_TILED=1 rate t64_tiled $V/libdtqn_hip_t64.so 2
DTQN_TRAIN_TILED=1 timeout 200 env DTQN_HIP_LIB=$V/libdtqn_hip_t64.so python -m pytest tests/test_gpu_td.py -q -p no:cacheprovider -k "batch_256" 2>&1 | tail -2
