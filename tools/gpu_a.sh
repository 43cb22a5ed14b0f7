#!/bin/bash
# round-2 GPU call A: full -m gpu suite (new parity-hole / surface tests), guard-fold probe, baseline bench
set -u
mkdir -p gpurun_out/a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/a/pytest.log
tail -15 gpurun_out/a/pytest.log
timeout 300 python tests/perf/guard_fold_probe.py dump gpurun_out/a/probe_default.npz 40 > gpurun_out/a/probe.log 2>&1
DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_fold.so timeout 300 python tests/perf/guard_fold_probe.py dump gpurun_out/a/probe_fold.npz 40 >> gpurun_out/a/probe.log 2>&1
python tests/perf/guard_fold_probe.py compare gpurun_out/a/probe_default.npz gpurun_out/a/probe_fold.npz >> gpurun_out/a/probe.log 2>&1
cat gpurun_out/a/probe.log
DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_fold.so timeout 300 python -m pytest tests/test_gpu_td.py -q -k "vs_oracle or reproducible" > gpurun_out/a/pytest_fold.log 2>&1
tail -5 gpurun_out/a/pytest_fold.log
timeout 600 python bench.py > gpurun_out/a/bench.json 2> gpurun_out/a/bench.err
cut -c1-1500 gpurun_out/a/bench.json
