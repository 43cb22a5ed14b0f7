#!/bin/bash
# round 5, GPU session 1 (one gpurun call): the whole -m gpu suite on the new engine (pipelined-parity cases, injected exchange failures,
# padded + action-embedding shapes), then the cfg-1 headline new vs the round-4 tree (tools/variants/_trees/base: same box, same minute),
# a kernel trace of both, the stage clocks of cfg 1 / cfg 2 (prof variant).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/s1; mkdir -p $OUT
T0=$SECONDS
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1
RC=$?
echo "tests rc=$RC t=$((SECONDS - T0))s"; tail -5 $OUT/tests.log | cut -c1-400
grep -E "^(FAILED|ERROR)" $OUT/tests.log | head -20
B="--no-other-configs --no-cpu-baseline --no-env-rate"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    print(sys.argv[1], round(d['value'], 1), 'upd/s', round(d['ms_per_step'] * 1e3, 2), 'us', d.get('kernels_us'))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
for R in 1 2; do
  timeout 90 python bench.py --steps 2000 --warmup 200 $B > $OUT/bench_new_2000_$R.json 2> $OUT/bench_new_2000_$R.err; show $OUT/bench_new_2000_$R.json
  (cd tools/variants/_trees/base && timeout 90 python bench.py --steps 2000 --warmup 200 $B > $GRAFT_REPO_ROOT/$OUT/bench_base_2000_$R.json 2> $GRAFT_REPO_ROOT/$OUT/bench_base_2000_$R.err); show $OUT/bench_base_2000_$R.json
done
timeout 90 python bench.py --steps 20 --warmup 5 $B > $OUT/bench_new_20.json 2> $OUT/bench_new_20.err; show $OUT/bench_new_20.json
DTQN_DSMALL=128 timeout 90 python bench.py --steps 2000 --warmup 200 $B > $OUT/bench_new_dsmall128.json 2>/dev/null; show $OUT/bench_new_dsmall128.json
echo "benches t=$((SECONDS - T0))s"
# kernel traces (per-kernel averages inside the pipeline)
for W in new base; do
  D=$GRAFT_REPO_ROOT; [ $W = base ] && D=$GRAFT_REPO_ROOT/tools/variants/_trees/base
  (cd $D && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/kt_$W -- python bench.py --steps 400 --warmup 40 $B > $GRAFT_REPO_ROOT/$OUT/bench_kt_$W.log 2>&1)
  DB=$(find $OUT/kt_$W -name '*results.db' | head -1)
  python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$W.md > /dev/null 2>&1
  rm -rf $OUT/kt_$W
  echo "== trace $W"; head -7 $OUT/kernel_stats_$W.md | cut -c1-200
done
echo "traces t=$((SECONDS - T0))s"
# stage clocks (prof variant of the NEW engine)
export DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_prof.so
timeout 60 python tests/perf/stage_profile.py 32 > $OUT/stage_cfg1.txt 2>&1; tail -34 $OUT/stage_cfg1.txt
DTQN_FWD_SLICES=4 timeout 60 python tests/perf/stage_profile.py 32 > $OUT/stage_cfg1_fwd4.txt 2>&1; sed -n 1,18p $OUT/stage_cfg1_fwd4.txt
timeout 60 python tests/perf/stage_profile.py 256 > $OUT/stage_cfg2.txt 2>&1; tail -34 $OUT/stage_cfg2.txt
echo "done t=$((SECONDS - T0))s"
