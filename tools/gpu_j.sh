#!/bin/bash
set -u
mkdir -p gpurun_out/j
B="python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-env-rate --no-other-configs"
for c in 1 2; do
  timeout 300 $B --config $c > gpurun_out/j/bench_cfg${c}.json 2>gpurun_out/j/bench_cfg${c}.err
  python -c "
import json
d=json.loads(open('gpurun_out/j/bench_cfg${c}.json').read().strip().splitlines()[-1])
print('cfg${c}:', round(d['value'],1), 'upd/s', {k: round(v,1) for k,v in d['kernels_us'].items()}, d['update_latency_us']['us_median'])" || tail -3 gpurun_out/j/bench_cfg${c}.err
done
