#!/bin/bash
set -u
mkdir -p gpurun_out/p
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_forward.py tests/test_gpu_parity_holes.py -m gpu -q --timeout 900 -x > gpurun_out/p/pytest.log 2>&1
tail -3 gpurun_out/p/pytest.log
B="python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-env-rate --no-other-configs"
for c in 3 1 2; do
  timeout 300 $B --config $c > gpurun_out/p/bench_cfg${c}.json 2>gpurun_out/p/bench_cfg${c}.err
  python -c "
import json
d=json.loads(open('gpurun_out/p/bench_cfg${c}.json').read().strip().splitlines()[-1])
print('cfg${c}:', round(d['value'],1), 'upd/s', round(d['ms_per_step'],3), 'ms', {k: round(v,1) for k,v in d['kernels_us'].items()})" || tail -3 gpurun_out/p/bench_cfg${c}.err
done
