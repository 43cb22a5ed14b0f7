#!/bin/bash
# round 6, session 13: full -m gpu suite on the side-stream chain / live-step weight gradients, then rates
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2300 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6
echo "== rates (default)"
python tests/perf/time_agent_cfg.py 3 4 5 2>&1 | grep cfg
python tests/perf/time_stages_cfg.py 3 4 5 --out gpurun_out/r06_s13_stages.json 2>&1 | grep cfg
