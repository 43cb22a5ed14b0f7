#!/bin/bash
# round 5, GPU session 5: A/B of the product build against tools/variants/libdtqn_hip_prev.so (the previous product): cfg-1 traces + rates, cfg 2 rate
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/s5; mkdir -p $OUT
T0=$SECONDS
B="--no-other-configs --no-cpu-baseline --no-env-rate"
P=$GRAFT_REPO_ROOT/dtqn_amd/csrc/libdtqn_hip.so
V=$GRAFT_REPO_ROOT/tools/variants
trace() {   # tag, lib, config, steps
  local W=$1 LIB=$2 C=$3 S=$4
  DTQN_HIP_LIB=$LIB timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_$W -- python bench.py --config $C --steps $S --warmup 50 $B > $OUT/bench_kt_$W.log 2>&1
  local DB=$(find $OUT/kt_$W -name '*results.db' | head -1)
  python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$W.md > /dev/null 2>&1
  rm -rf $OUT/kt_$W
  echo "== $W"; head -6 $OUT/kernel_stats_$W.md | tail -4 | awk -F'|' '{print substr($2,1,40), $3, $5, $6}'
}
rate() {   # tag, lib, config, steps, warmup
  DTQN_HIP_LIB=$2 timeout 120 python bench.py --config $3 --steps $4 --warmup $5 $B 2>/dev/null > $OUT/rate_$1.json
  python -c "
import json,sys
d=json.loads([l for l in open('$OUT/rate_$1.json') if l.startswith('{')][-1]); print('rate $1', round(d['value'],1), 'upd/s', round(d['ms_per_step']*1e3,2), 'us', d.get('kernels_us'))"
}
if [ "${TESTS:-1}" = 1 ]; then
timeout 400 python -m pytest tests/test_gpu_td.py tests/test_gpu_pipelined_parity.py tests/test_gpu_pipeline.py tests/test_gpu_agent.py tests/test_gpu_dp.py -q -p no:cacheprovider -x > $OUT/tests.log 2>&1
echo "tests rc=$? t=$((SECONDS - T0))s"; tail -3 $OUT/tests.log | cut -c1-300; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
fi
trace prod $P 1 600
trace prev $V/libdtqn_hip_prev.so 1 600
for R in 1 2; do
  rate prod_2000_$R $P 1 2000 200
  rate prev_2000_$R $V/libdtqn_hip_prev.so 1 2000 200
done
rate prod_20 $P 1 20 5
rate prod_cfg2 $P 2 400 50
rate prev_cfg2 $V/libdtqn_hip_prev.so 2 400 50
echo "done t=$((SECONDS - T0))s"
