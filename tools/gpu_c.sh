#!/bin/bash
# round-2 GPU call C: weights-through-LDS forward: parity + A/B timing
set -u
mkdir -p gpurun_out/c
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_td.py tests/test_gpu_parity_holes.py -m gpu -q -x --timeout 600 > gpurun_out/c/pytest.log 2>&1
tail -5 gpurun_out/c/pytest.log
B="python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-env-rate --no-other-configs"
for wl in 0 1; do
  DTQN_WL=$wl timeout 300 $B > gpurun_out/c/bench_cfg1_wl$wl.json 2>gpurun_out/c/bench_cfg1_wl$wl.err
  DTQN_WL=$wl timeout 300 $B --config 2 > gpurun_out/c/bench_cfg2_wl$wl.json 2>gpurun_out/c/bench_cfg2_wl$wl.err
done
python - <<'PY'
import json
for c in (1, 2):
    for wl in (0, 1):
        try:
            d = json.loads(open(f"gpurun_out/c/bench_cfg{c}_wl{wl}.json").read().strip().splitlines()[-1])
            print(f"cfg{c} wl={wl}: {d['value']:.0f} upd/s, {d['ms_per_step']*1e3:.1f} us/update, kernels {json.dumps({k: round(v, 1) for k, v in d['kernels_us'].items()})}")
        except Exception as e:
            print(c, wl, "failed", e)
PY
