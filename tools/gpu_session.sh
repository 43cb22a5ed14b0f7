cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2
timeout 300 python tests/perf/time_fwd_parts.py > gpurun_out/s2/fwd_parts.txt 2>&1
cat gpurun_out/s2/fwd_parts.txt
DTQN_HIP_LIB=tools/variants/libdtqn_hip_prof4.so DTQN_FWD_SLICES=4 timeout 300 python tests/perf/stage_profile.py > gpurun_out/s2/stage_rs4.txt 2>&1
DTQN_HIP_LIB=tools/variants/libdtqn_hip_prof4.so timeout 300 python tests/perf/stage_profile.py > gpurun_out/s2/stage_rs2.txt 2>&1
paste gpurun_out/s2/stage_rs2.txt gpurun_out/s2/stage_rs4.txt | head -20
DTQN_FWD_SLICES=4 timeout 600 python -m pytest tests/test_gpu_td.py -q -m gpu -x > gpurun_out/s2/td_rs4.log 2>&1
tail -3 gpurun_out/s2/td_rs4.log
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/s2/all_tests.log 2>&1
echo "all tests rc=$?" >> gpurun_out/s2/all_tests.log
tail -8 gpurun_out/s2/all_tests.log
