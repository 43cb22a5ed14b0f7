cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s19
for v in product l2scope product l2scope; do
  if [ $v = product ]; then unset DTQN_HIP_LIB; else export DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_$v.so; fi
  timeout 300 python bench.py --steps 2000 --warmup 200 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s19/bench_$v.json 2> gpurun_out/s19/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/s19/bench_$v.json') if l.startswith('{')][0])
    print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'])
except Exception as e:
    print('$v failed', e); print(open('gpurun_out/s19/bench_$v.err').read()[-800:])
PY
done
export DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_l2scope.so
timeout 600 python -m pytest tests/test_gpu_td.py tests/test_gpu_pipeline.py tests/test_gpu_loop_golden.py -q -m gpu -x > gpurun_out/s19/tests_l2.log 2>&1
tail -4 gpurun_out/s19/tests_l2.log
