#!/bin/bash
set -u
mkdir -p gpurun_out/e
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_td.py -m gpu -q -x --timeout 600 > gpurun_out/e/pytest.log 2>&1
tail -3 gpurun_out/e/pytest.log
DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_prof.so timeout 200 python tests/perf/stage_profile.py 32 > gpurun_out/e/stage_profile_B32.log 2>&1
cat gpurun_out/e/stage_profile_B32.log
B="python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-env-rate --no-other-configs"
for c in 1 2 3; do
  timeout 300 $B --config $c > gpurun_out/e/bench_cfg$c.json 2>gpurun_out/e/bench_cfg$c.err
done
python - <<'PY'
import json
for c in (1, 2, 3):
    try:
        d = json.loads(open(f"gpurun_out/e/bench_cfg{c}.json").read().strip().splitlines()[-1])
        print(f"cfg{c}: {d['value']:.0f} upd/s, {d['ms_per_step']*1e3:.1f} us/update, kernels {json.dumps({k: round(v, 1) for k, v in d['kernels_us'].items()})}")
    except Exception as e:
        print(c, "failed", e)
PY
