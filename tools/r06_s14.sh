#!/bin/bash
# round 6, session 14: small weight-gradient jobs on the fork-join lane; embedding product table
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_full_size.py tests/test_gpu_pipelined_parity.py tests/test_gpu_dp.py tests/test_gpu_forward.py -x -q -p no:cacheprovider 2>&1 | tail -4
echo "== rates (default: lane + table)"
python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates DTQN_WGRAD_SIDE=0"
DTQN_WGRAD_SIDE=0 python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates DTQN_EMBED_TABLE=0"
DTQN_EMBED_TABLE=0 python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
python tests/perf/time_stages_cfg.py 3 4 5 2>&1 | grep cfg
