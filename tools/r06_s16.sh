#!/bin/bash
# round 6, session 16: next layer's q|k|v projection inside the fused layer launch
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_full_size.py tests/test_gpu_pipelined_parity.py tests/test_gpu_forward.py -x -q -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
echo "== rates (default: qkv tail)"; python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates DTQN_QKV_FUSE=0"; DTQN_QKV_FUSE=0 python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
done
for T in 1000 2000 3000; do echo "== DTQN_SKEW_LAYER=$T"; DTQN_SKEW_LAYER=$T python tests/perf/time_stages_cfg.py 4 2>&1 | grep cfg; done
