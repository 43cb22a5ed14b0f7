#!/bin/bash
# round 6, session 12: D = 256 chain, Q-head backward in the dx staging, per-kernel skew: parity, then rates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_s12
echo "== parity (row-block cases)"
timeout 1500 python -m pytest tests/test_gpu_td.py tests/test_gpu_full_size.py tests/test_gpu_forward.py tests/test_gpu_pipelined_parity.py -x -q -p no:cacheprovider 2>&1 | tail -4
echo "== parity with DTQN_BWD_CHAIN256=1"
DTQN_BWD_CHAIN256=1 timeout 1500 python -m pytest tests/test_gpu_td.py tests/test_gpu_full_size.py tests/test_gpu_pipelined_parity.py -x -q -p no:cacheprovider -k "cfg5 or 256 or config5 or hallway" 2>&1 | tail -4
echo "== rates (default)"
python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates DTQN_BWD_CHAIN256=1"
DTQN_BWD_CHAIN256=1 python tests/perf/time_agent_cfg.py 5 2>&1 | grep cfg
echo "== rates DTQN_HEAD_FUSE=0"
DTQN_HEAD_FUSE=0 python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== stages (default)"
python tests/perf/time_stages_cfg.py 3 4 5 --out gpurun_out/r06_s12/stages.json 2>&1 | grep cfg
echo "== stages DTQN_BWD_CHAIN256=1"
DTQN_BWD_CHAIN256=1 python tests/perf/time_stages_cfg.py 5 2>&1 | grep cfg
echo "== stages cfg5 DTQN_BWD_CHAIN256=1 DTQN_PIPELINE=0 agent"
DTQN_BWD_CHAIN256=1 DTQN_PIPELINE=0 python tests/perf/time_agent_cfg.py 5 2>&1 | grep cfg
for T in 300 600 1000; do echo "== DTQN_SKEW_WIDE=$T"; DTQN_SKEW_WIDE=$T python tests/perf/time_stages_cfg.py 4 2>&1 | grep cfg; done
bash tools/r06_trace.sh 3 cfg3_fused | head -32
bash tools/r06_trace.sh 5 cfg5_fused | head -32
