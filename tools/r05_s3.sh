#!/bin/bash
# round 5, GPU session 3: the engine with the options that paid (DTQN_OPT = 4 | 32 | 64 | 128) against the same sources with all of them
# off (n0) and without the embedding hoist (n100): kernel traces at cfg 1, rates at cfg 1 (2000 / 20 steps) and cfg 2, the -m gpu suite.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/s3; mkdir -p $OUT
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
  echo "== $W"; head -8 $OUT/kernel_stats_$W.md | tail -6 | cut -c1-60,96-160
}
rate() {   # tag, lib, config, steps, warmup
  DTQN_HIP_LIB=$2 timeout 120 python bench.py --config $3 --steps $4 --warmup $5 $B 2>/dev/null > $OUT/rate_$1.json
  python -c "
import json,sys
d=json.loads([l for l in open('$OUT/rate_$1.json') if l.startswith('{')][-1]); print('rate $1', round(d['value'],1), 'upd/s', round(d['ms_per_step']*1e3,2), 'us', d.get('kernels_us'))"
}
trace prod_cfg1 $P 1 600
trace n0_cfg1 $V/libdtqn_hip_n0.so 1 600
trace n100_cfg1 $V/libdtqn_hip_n100.so 1 600
echo "traces t=$((SECONDS - T0))s"
for R in 1 2; do
  rate prod_2000_$R $P 1 2000 200
  rate n0_2000_$R $V/libdtqn_hip_n0.so 1 2000 200
done
rate prod_20 $P 1 20 5
HIP_FORCE_DEV_KERNARG=1 rate prod_devkernarg1 $P 1 2000 200
HIP_FORCE_DEV_KERNARG=0 rate prod_devkernarg0 $P 1 2000 200
rate prod_cfg2 $P 2 400 50
rate n0_cfg2 $V/libdtqn_hip_n0.so 2 400 50
rate n100_cfg2 $V/libdtqn_hip_n100.so 2 400 50
trace prod_cfg2 $P 2 200
echo "rates t=$((SECONDS - T0))s"
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? t=$((SECONDS - T0))s"; tail -4 $OUT/tests.log | cut -c1-300; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head -20
echo "done t=$((SECONDS - T0))s"
