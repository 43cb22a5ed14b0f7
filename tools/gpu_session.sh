cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s8
for v in product dsmall256 product dsmall256; do
  if [ $v = product ]; then unset DTQN_HIP_LIB; else export DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_$v.so; fi
  timeout 300 python bench.py --steps 2000 --warmup 200 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s8/bench_$v.json 2> gpurun_out/s8/bench_$v.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/s8/bench_$v.json') if l.startswith('{')][0])
print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'], d['hbm_kernels']['replay_apply'])
PY
done
unset DTQN_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity_holes.py tests/test_gpu_pipeline.py tests/test_gpu_agent.py -q -m gpu > gpurun_out/s8/tests.log 2>&1
tail -3 gpurun_out/s8/tests.log
timeout 1500 bash tools/learning_curves.sh "overlap:1 refq:1 overlap:2 refq:2 overlap:3 refq:3" > gpurun_out/s8/curves.log 2>&1
tail -12 gpurun_out/s8/curves.log
