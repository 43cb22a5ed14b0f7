#!/bin/bash
# round 5, GPU session 4: write-through record stores on the row-block path (DTQN_OPT bit 512) against the same sources without it (n228):
# BASELINE configs 3 / 4 / 5 and the image config's rates, the row-block GPU tests.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/s4; mkdir -p $OUT
T0=$SECONDS
B="--no-other-configs --no-cpu-baseline --no-env-rate"
P=$GRAFT_REPO_ROOT/dtqn_amd/csrc/libdtqn_hip.so
V=$GRAFT_REPO_ROOT/tools/variants
rate() {   # tag, lib, config, steps, warmup
  DTQN_HIP_LIB=$2 timeout 200 python bench.py --config $3 --steps $4 --warmup $5 $B 2>$OUT/rate_$1.err > $OUT/rate_$1.json
  python -c "
import json,sys
try:
    d=json.loads([l for l in open('$OUT/rate_$1.json') if l.startswith('{')][-1]); print('rate $1', round(d['value'],1), 'upd/s', round(d['ms_per_step']*1e3,2), 'us')
except Exception as e: print('rate $1 failed', e); print(open('$OUT/rate_$1.err').read()[-800:])"
}
timeout 400 python -m pytest tests/test_gpu_td.py tests/test_gpu_forward.py tests/test_gpu_bag.py tests/test_gpu_parity_holes.py -q -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? t=$((SECONDS - T0))s"; tail -3 $OUT/tests.log | cut -c1-300; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
for C in 3 4 5; do
  S=150; [ $C = 5 ] && S=100
  rate prod_cfg$C $P $C $S 20
  rate n228_cfg$C $V/libdtqn_hip_n228.so $C $S 20
done
rate prod_cfg3b $P 3 150 20
rate n228_cfg3b $V/libdtqn_hip_n228.so 3 150 20
echo "done t=$((SECONDS - T0))s"
