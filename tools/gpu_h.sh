#!/bin/bash
set -u
mkdir -p gpurun_out/h
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_parity_holes.py tests/test_gpu_agent.py -m gpu -q -x --timeout 600 > gpurun_out/h/pytest.log 2>&1
tail -5 gpurun_out/h/pytest.log
B="python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-env-rate --no-other-configs"
for wl in f 1; do
  for c in 1 2; do
    DTQN_WL=$wl timeout 300 $B --config $c > gpurun_out/h/bench_cfg${c}_wl$wl.json 2>gpurun_out/h/bench_cfg${c}_wl$wl.err
    python -c "
import json
d=json.loads(open('gpurun_out/h/bench_cfg${c}_wl$wl.json').read().strip().splitlines()[-1])
print('cfg${c} wl=$wl:', round(d['value'],1), 'upd/s', {k: round(v,1) for k,v in d['kernels_us'].items()}, d['update_latency_us']['us_median'])" || tail -3 gpurun_out/h/bench_cfg${c}_wl$wl.err
  done
done
