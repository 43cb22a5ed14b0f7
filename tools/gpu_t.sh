#!/bin/bash
# tiled GEMM epilogue / fragment experiment: parity of the tiled path, cfg 4/5 rates, kernel trace of cfg 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_bag.py tests/test_gpu_forward.py -x -q -m gpu > gpurun_out/t_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t_tests.log
tail -4 gpurun_out/t_tests.log
for c in 4 5; do
  timeout 300 python bench.py --config $c --steps 300 --warmup 30 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/t_bench_cfg$c.json 2> gpurun_out/t_bench_cfg$c.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/t_bench_cfg$c.json') if l.startswith('{')][0])
print('cfg$c', d['value'], d['ms_per_step'], d['roofline'].get('frac'))
PY
done
export TMPDIR=/tmp
rm -rf /tmp/kt4
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- python bench.py --config 4 --steps 100 --warmup 20 --no-cpu-baseline --no-env-rate --no-other-configs > /dev/null 2>&1
DB=$(find /tmp/kt4 -name '*results.db' | head -1)
python tools/rocpd_summary.py "$DB" gpurun_out/t_kernel_stats_cfg4.md > /dev/null
head -8 gpurun_out/t_kernel_stats_cfg4.md
