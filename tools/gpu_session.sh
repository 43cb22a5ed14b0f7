#!/bin/bash
# end-of-round-4 session (one gpurun call): the whole -m gpu suite and smoke on the final engine; if green, the cfg-1 trace + counters that
# bench.py reports (profiles/r04e_*: stamped with the engine digest), the headline at 2 000 and at the driver's 20 steps, and (FULL_BENCH=1) the default line.
#   gpurun --timeout 700 -- "GIT_HEAD=<short sha> bash tools/gpu_session.sh"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fin
T0=$SECONDS
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/fin/tests.log 2>&1
RC=$?
echo "tests rc=$RC t=$((SECONDS - T0))s"; tail -6 gpurun_out/fin/tests.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/fin/tests.log | head -20
timeout 90 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
[ $RC -ne 0 ] && exit 0
GIT_HEAD=${GIT_HEAD:-unknown} timeout 280 bash tools/profile_round4.sh r04e "1" > gpurun_out/fin/profile.log 2>&1
echo "profile t=$((SECONDS - T0))s"; ls gpurun_out/r04e/cfg1/
for K in "2000 200" "20 5"; do
  set -- $K
  timeout 90 python bench.py --steps $1 --warmup $2 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/fin/bench_$1.json 2> gpurun_out/fin/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/fin/bench_$1.json') if l.startswith('{')][0])
    print('steps $1:', round(d['value'], 1), round(d['ms_per_step'] * 1e3, 2), d['kernels_us'], d['roofline']['traffic'], d['roofline']['traffic_src'])
except Exception as e:
    print('bench $1 failed', e); print(open('gpurun_out/fin/bench_$1.err').read()[-600:])
PY
done
echo "quick benches t=$((SECONDS - T0))s"
for C in ${OTHER_CFGS:-}; do      # row-block configs (their attention kernels take the softmax scale as an argument since the head padding)
  timeout 60 python bench.py --config $C --steps 200 --warmup 30 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/fin/bench_cfg$C.json 2> gpurun_out/fin/bench_cfg$C.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/fin/bench_cfg$C.json') if l.startswith('{')][0])
    print('cfg $C:', round(d['value'], 1), round(d['ms_per_step'], 4))
except Exception as e:
    print('cfg $C failed', e)
PY
done
if [ "${FULL_BENCH:-0}" = 1 ]; then
timeout 260 python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/fin/bench_default.json') if l.startswith('{')][0])
    print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('cpu_baseline', {}).get('value'))
    print({k: v.get('updates_per_s') for k, v in d.get('other_configs', {}).items()}, d.get('env_steps_per_s'))
except Exception as e:
    print('default bench failed', e); print(open('gpurun_out/fin/bench_default.err').read()[-600:])
PY
fi
echo "done t=$((SECONDS - T0))s"
