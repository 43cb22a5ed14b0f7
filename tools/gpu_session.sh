cd "$GRAFT_REPO_ROOT" || exit 1
GIT_HEAD=07d60d6 bash tools/profile_round4.sh r04b "1 5" > gpurun_out/r04b_profile.log 2>&1
tail -3 gpurun_out/r04b_profile.log
mkdir -p gpurun_out/s18
timeout 300 python -m pytest tests/test_gpu_dp.py tests/test_gpu_agent.py tests/test_gpu_pipeline.py -q -m gpu > gpurun_out/s18/tests.log 2>&1
tail -3 gpurun_out/s18/tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s18/bench_full.json 2> gpurun_out/s18/bench_full.err
python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/s18/bench_full.json') if l.startswith('{')][0])
    print(round(d['value'],1), round(d['ms_per_step']*1e3,2), d['roofline']['traffic_src'], d['other_configs'], d['env_steps_per_sec'])
except Exception as e:
    print('failed', e); print(open('gpurun_out/s18/bench_full.err').read()[-1500:])
PY
