cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s9
timeout 200 python tests/perf/overlap_loop_rate.py 3 2>&1 | tail -1
DTQN_TIMING_ONLY_NO_ACTOR_WAIT=1 timeout 200 python tests/perf/overlap_loop_rate.py 3 2>&1 | tail -1
timeout 200 python tests/perf/overlap_loop_rate.py 3 2>&1 | tail -1
DTQN_TIMING_ONLY_NO_ACTOR_WAIT=1 timeout 200 python tests/perf/overlap_loop_rate.py 3 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_bench_contract.py tests/test_gpu_agent.py tests/test_gpu_pipeline.py -q -m gpu > gpurun_out/s9/tests.log 2>&1
tail -3 gpurun_out/s9/tests.log
