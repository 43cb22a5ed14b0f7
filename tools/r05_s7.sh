#!/bin/bash
# round 5, GPU session 7: BASELINE config 2 experiments: half weight fragments in time in the backward's dh W_1 (halfw), matrix-core attention at head_dim 8
# on whole-sequence workgroups (mfma8), against the product build (prev)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/s7; mkdir -p $OUT
B="--no-other-configs --no-cpu-baseline --no-env-rate"
V=$GRAFT_REPO_ROOT/tools/variants
rate() {   # tag lib config
  DTQN_HIP_LIB=$2 timeout 120 python bench.py --config $3 --steps 300 --warmup 40 $B 2>$OUT/rate_$1.err > $OUT/rate_$1.json
  python -c "
import json,sys
try:
    d=json.loads([l for l in open('$OUT/rate_$1.json') if l.startswith('{')][-1]); print('rate $1', round(d['value'],1), 'upd/s', round(d['ms_per_step']*1e3,2), 'us', d.get('kernels_us'))
except Exception as e: print('rate $1 failed', e, open('$OUT/rate_$1.err').read()[-400:])"
}
for R in 1 2; do
rate prev_$R $V/libdtqn_hip_prev.so 2
rate halfw_$R $V/libdtqn_hip_halfw.so 2
rate mfma8_$R $V/libdtqn_hip_mfma8.so 2
done
rate mfma8_cfg1 $V/libdtqn_hip_mfma8.so 1
rate prev_cfg1 $V/libdtqn_hip_prev.so 1
timeout 200 env DTQN_HIP_LIB=$V/libdtqn_hip_mfma8.so python -m pytest tests/test_gpu_td.py -q -p no:cacheprovider -k "batch_256 or kw2 or kw3" 2>&1 | tail -2
timeout 200 env DTQN_HIP_LIB=$V/libdtqn_hip_halfw.so python -m pytest tests/test_gpu_td.py -q -p no:cacheprovider -k "batch_256 or kw2 or kw3 or kw6" 2>&1 | tail -2
