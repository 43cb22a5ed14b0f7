#!/bin/bash
# round 5 closing session (one gpurun call): the whole -m gpu suite and smoke on the final engine; if green, the cfg-1 kernel trace + counter
# passes bench.py reports (profiles/r05_*: stamped with the engine digest), the headline at 2 000 and at the driver's 20 steps, the stage
# clocks, a cfg-2 trace, and the default bench line.
#   gpurun --timeout 1200 -- "GIT_HEAD=<short sha> bash tools/r05_final.sh"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/fin
T0=$SECONDS
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/fin/tests.log 2>&1
RC=$?
echo "tests rc=$RC t=$((SECONDS - T0))s"; tail -5 gpurun_out/fin/tests.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/fin/tests.log | head -20
timeout 90 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
[ $RC -ne 0 ] && exit 0
GIT_HEAD=${GIT_HEAD:-unknown} timeout 400 bash tools/profile_round4.sh r05 "1" > gpurun_out/fin/profile.log 2>&1
echo "profile t=$((SECONDS - T0))s"; ls gpurun_out/r05/cfg1/; head -8 gpurun_out/r05/cfg1/kernel_stats.md | cut -c1-60,96-170
B="--no-other-configs --no-env-rate --no-cpu-baseline"
for K in "2000 200" "20 5" "20 5"; do
  set -- $K
  timeout 90 python bench.py --steps $1 --warmup $2 $B > gpurun_out/fin/bench_$1.json 2> gpurun_out/fin/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/fin/bench_$1.json') if l.startswith('{')][0])
    print('steps $1:', round(d['value'], 1), round(d['ms_per_step'] * 1e3, 2), d['kernels_us'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_src'], d.get('pipeline'))
except Exception as e:
    print('bench $1 failed', e); print(open('gpurun_out/fin/bench_$1.err').read()[-600:])
PY
done
echo "quick benches t=$((SECONDS - T0))s"
# head width 32 / width-padded shapes: rates (one process per shape) and the kernel trace of two of them
timeout 240 python tests/perf/padded_rate.py > gpurun_out/fin/padded_shapes_rate.txt 2>&1; cat gpurun_out/fin/padded_shapes_rate.txt | cut -c1-150
for S in "64 2" "48 6"; do
  set -- $S
  rocprofv3 --kernel-trace --stats -d gpurun_out/fin/ktp -- python tests/perf/padded_rate.py $1 $2 0 > gpurun_out/fin/ktp_$1_$2.log 2>&1
  python tools/rocpd_summary.py "$(find gpurun_out/fin/ktp -name '*results.db' | head -1)" gpurun_out/fin/kernel_stats_inembed$1_heads$2.md > /dev/null 2>&1; rm -rf gpurun_out/fin/ktp
  head -6 gpurun_out/fin/kernel_stats_inembed$1_heads$2.md | cut -c1-75,96-170
done
echo "padded shapes t=$((SECONDS - T0))s"
# batches past latency mode (sliced backward under whole-sequence forward passes): rate and kernel trace at 64 and 128 sequences
for BS in 64 128; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/fin/ktb -- python bench.py --config 2 --batch $BS --steps 400 --warmup 60 $B > gpurun_out/fin/bench_batch$BS.json 2> gpurun_out/fin/bench_batch$BS.err
  python tools/rocpd_summary.py "$(find gpurun_out/fin/ktb -name '*results.db' | head -1)" gpurun_out/fin/kernel_stats_batch$BS.md > /dev/null 2>&1; rm -rf gpurun_out/fin/ktb
  head -6 gpurun_out/fin/kernel_stats_batch$BS.md | cut -c1-75,96-170
  python -c "
import json
d = json.loads([l for l in open('gpurun_out/fin/bench_batch$BS.json') if l.startswith('{')][0]); print('batch $BS (under the tracer):', round(d['value'], 1), 'upd/s')"
done
DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_prof.so timeout 60 python tests/perf/stage_profile.py 32 > gpurun_out/fin/stage_cfg1.txt 2>&1
DTQN_FWD_SLICES=4 DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_prof.so timeout 60 python tests/perf/stage_profile.py 32 > gpurun_out/fin/stage_cfg1_fwd4.txt 2>&1
DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_prof.so timeout 60 python tests/perf/stage_profile.py 256 > gpurun_out/fin/stage_cfg2.txt 2>&1
tail -34 gpurun_out/fin/stage_cfg1.txt | head -18
rocprofv3 --kernel-trace --stats -d gpurun_out/fin/kt2 -- python bench.py --config 2 --steps 200 --warmup 40 $B > gpurun_out/fin/bench_kt2.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/fin/kt2 -name '*results.db' | head -1)" gpurun_out/fin/kernel_stats_cfg2.md > /dev/null 2>&1; rm -rf gpurun_out/fin/kt2
head -7 gpurun_out/fin/kernel_stats_cfg2.md | cut -c1-60,96-170
echo "profiles t=$((SECONDS - T0))s"
if [ "${FULL_BENCH:-1}" = 1 ]; then
timeout 400 python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/fin/bench_default.json') if l.startswith('{')][0])
    print('default:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('cpu_baseline', {}).get('value'))
    print({k: (v.get('updates_per_s'), v.get('frac_mfma')) for k, v in d.get('other_configs', {}).items()}, d.get('env_steps_per_sec'))
    print(len(open('gpurun_out/fin/bench_default.json').read()), 'bytes')
except Exception as e:
    print('default bench failed', e); print(open('gpurun_out/fin/bench_default.err').read()[-600:])
PY
fi
echo "done t=$((SECONDS - T0))s"
