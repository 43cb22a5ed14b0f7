#!/bin/bash
# round 2: bag feature on the device + full gpu suite + bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bag.py -x -q -m gpu > gpurun_out/q_bag.log 2>&1
echo "bag rc=$?" >> gpurun_out/q_bag.log
tail -15 gpurun_out/q_bag.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_bag.py > gpurun_out/q_all.log 2>&1
echo "all rc=$?" >> gpurun_out/q_all.log
tail -8 gpurun_out/q_all.log
timeout 600 python bench.py > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err
tail -c 1500 gpurun_out/q_bench.json
