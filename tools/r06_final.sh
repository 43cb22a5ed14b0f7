#!/bin/bash
# round 6 closing session: whole -m gpu suite, smoke, then trace + counter passes of every BASELINE config on this engine
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 2300 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r06/z_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06/z_tests.log
tail -4 gpurun_out/r06/z_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/profile_round4.sh r06 "${CFGS:-1 2 3 4 5}" > gpurun_out/r06/z_profile.log 2>&1
tail -5 gpurun_out/r06/z_profile.log
python tests/perf/time_agent_cfg.py 2 3 4 5 2>&1 | grep cfg | tee gpurun_out/r06/z_rates.txt
