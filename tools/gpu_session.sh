cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s17
timeout 700 bash tools/dp_modes_one_gpu.sh 2>&1 | grep -E "^OK|^FAIL"
timeout 300 python bench.py --steps 2000 --warmup 200 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s17/bench2000.json 2> gpurun_out/s17/bench2000.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s17/bench20.json 2> gpurun_out/s17/bench20.err
python - <<PY
import json
for f in ('bench2000','bench20'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/s17/{f}.json') if l.startswith('{')][0])
        print(f, round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'])
    except Exception as e:
        print(f, 'failed', e); print(open(f'gpurun_out/s17/{f}.err').read()[-1500:])
PY
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s17/all_tests.log 2>&1
echo "all tests rc=$?" >> gpurun_out/s17/all_tests.log
tail -5 gpurun_out/s17/all_tests.log
