#!/bin/bash
# round 6, session 10: fused layer tail (tl_layer_kernel) and backward chain (tl_chain_bwd_kernel): parity first, then A/B rates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_s10
echo "== parity (row-block cases)"
timeout 1200 python -m pytest tests/test_gpu_td.py tests/test_gpu_full_size.py tests/test_gpu_forward.py tests/test_gpu_pipelined_parity.py -x -q -p no:cacheprovider 2>&1 | tail -4
echo "== rates: fused (default)"
python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates: DTQN_LAYER_FUSE=0"
DTQN_LAYER_FUSE=0 python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== rates: DTQN_BWD_CHAIN=0"
DTQN_BWD_CHAIN=0 python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
echo "== stages fused"
python tests/perf/time_stages_cfg.py 3 4 5 --out gpurun_out/r06_s10/stages_fused.json 2>&1 | grep cfg
echo "== stages unfused"
DTQN_LAYER_FUSE=0 DTQN_BWD_CHAIN=0 python tests/perf/time_stages_cfg.py 3 4 5 --out gpurun_out/r06_s10/stages_unfused.json 2>&1 | grep cfg
bash tools/r06_trace.sh 4 cfg4_fused | head -30
