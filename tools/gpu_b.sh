#!/bin/bash
# round-2 GPU call B: (1) was the "GPU-only wrong gradient" of the folded backward item guard the pad-row bug fixed in 5c94a3c?
#   trees at bc2e408 (before the fix) and 5c94a3c (with it), each with and without the fold; (2) failed test re-run
set -u
mkdir -p gpurun_out/b
R=$PWD
for d in prefix_plain prefix_fold postfix_plain postfix_fold; do
  (cd tools/variants/_trees/$d && timeout 200 python tests/perf/guard_fold_probe.py dump $R/gpurun_out/b/probe_$d.npz 20 > $R/gpurun_out/b/probe_$d.log 2>&1; tail -2 $R/gpurun_out/b/probe_$d.log)
done
python tests/perf/guard_fold_probe.py compare gpurun_out/b/probe_prefix_plain.npz gpurun_out/b/probe_prefix_fold.npz > gpurun_out/b/compare_prefix.log 2>&1
python tests/perf/guard_fold_probe.py compare gpurun_out/b/probe_postfix_plain.npz gpurun_out/b/probe_postfix_fold.npz > gpurun_out/b/compare_postfix.log 2>&1
python tests/perf/guard_fold_probe.py compare gpurun_out/b/probe_prefix_plain.npz gpurun_out/b/probe_postfix_plain.npz > gpurun_out/b/compare_plain_pre_post.log 2>&1
cat gpurun_out/b/compare_prefix.log gpurun_out/b/compare_postfix.log gpurun_out/b/compare_plain_pre_post.log
timeout 600 python -m pytest tests/test_gpu_parity_holes.py tests/test_gpu_surface.py tests/test_gpu_dp.py tests/test_bench_contract.py -m gpu -q --timeout 600 > gpurun_out/b/pytest.log 2>&1
tail -30 gpurun_out/b/pytest.log
DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_prof.so timeout 200 python tests/perf/stage_profile.py 32 > gpurun_out/b/stage_profile_B32.log 2>&1
cat gpurun_out/b/stage_profile_B32.log
