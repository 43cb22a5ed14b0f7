#!/bin/bash
# round 6, session 15: packed rows of the unsaved passes (config 3)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_full_size.py tests/test_gpu_pipelined_parity.py tests/test_gpu_forward.py -x -q -p no:cacheprovider 2>&1 | tail -4
echo "== rates (default: packed)"
python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates DTQN_PACK_ROWS=0"
DTQN_PACK_ROWS=0 python tests/perf/time_agent_cfg.py 3 2>&1 | grep cfg
python tests/perf/time_stages_cfg.py 3 2>&1 | grep cfg
for T in 0 1000 2000 3000; do echo "== DTQN_SKEW_LAYER=$T DTQN_SKEW_WIDE=$((T/3))"; DTQN_SKEW_LAYER=$T DTQN_SKEW_WIDE=$((T/3)) python tests/perf/time_stages_cfg.py 3 2>&1 | grep cfg; done
bash tools/r06_trace.sh 3 cfg3_packed | head -24
