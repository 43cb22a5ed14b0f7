#!/bin/bash
# round 2: tiled-path kernel rewrites (embed fwd/bwd on MFMA, LN backward per row block, qhead) -- parity, then cfg 4/5 rates + traces
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_td.py tests/test_gpu_bag.py tests/test_gpu_forward.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r_tests.log
tail -6 gpurun_out/r_tests.log
for c in 4 5; do
  timeout 300 python bench.py --config $c --steps 300 --warmup 30 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/r_bench_cfg$c.json 2> gpurun_out/r_bench_cfg$c.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r_bench_cfg$c.json') if l.startswith('{')][0])
print('cfg$c', d['value'], d['ms_per_step'], d['roofline'].get('frac'))
PY
done
bash tools/profile_round2.sh r02b "4 5" > gpurun_out/r_profile.log 2>&1
head -16 gpurun_out/r02b/cfg4/kernel_stats.md
head -14 gpurun_out/r02b/cfg5/kernel_stats.md
