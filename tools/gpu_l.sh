#!/bin/bash
set -u
mkdir -p gpurun_out/l
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/l/pytest.log 2>&1
tail -12 gpurun_out/l/pytest.log
B="python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-other-configs"
timeout 300 $B > gpurun_out/l/bench_cfg1.json 2>gpurun_out/l/bench_cfg1.err
python -c "
import json
d=json.loads(open('gpurun_out/l/bench_cfg1.json').read().strip().splitlines()[-1])
print('cfg1:', round(d['value'],1), 'upd/s', {k: round(v,1) for k,v in d['kernels_us'].items()}, d['update_latency_us']['us_median'])
print({k: (round(v,1) if isinstance(v,float) else v) for k,v in d['env_steps_per_sec'].items() if k!='note'})" || tail -3 gpurun_out/l/bench_cfg1.err
