#!/bin/bash
set -u
mkdir -p gpurun_out/n
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_forward.py -m gpu -q --timeout 900 -x -k "tiled or cfg345 or kw9 or kw10 or kw11 or kw12 or kw13 or kw14 or kw15 or kw16 or kw17 or variants" > gpurun_out/n/pytest.log 2>&1
tail -3 gpurun_out/n/pytest.log
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-env-rate --no-other-configs"
for v in product lb2; do
for c in 4 5 3; do
  if [ $v = lb2 ]; then export DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_lb2.so; else unset DTQN_HIP_LIB; fi
  timeout 300 $B --config $c > gpurun_out/n/bench_cfg${c}_$v.json 2>gpurun_out/n/bench_cfg${c}_$v.err
  python -c "
import json
d=json.loads(open('gpurun_out/n/bench_cfg${c}_$v.json').read().strip().splitlines()[-1])
print('cfg${c} $v:', round(d['value'],1), 'upd/s', round(d['ms_per_step'],3), 'ms')" || tail -3 gpurun_out/n/bench_cfg${c}_$v.err
done
done
