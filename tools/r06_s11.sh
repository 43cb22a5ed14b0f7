#!/bin/bash
# round 6, session 11: start skew / rows per workgroup of the fused layer kernel at config 4
cd "$GRAFT_REPO_ROOT" || exit 1
for T in 0 300 600 1200 2400 4000; do echo "== DTQN_SKEW_TICKS=$T"; DTQN_SKEW_TICKS=$T python tests/perf/time_stages_cfg.py 4 2>&1 | grep cfg; done
echo "== DTQN_ROWS_FFN=32"; DTQN_ROWS_FFN=32 python tests/perf/time_stages_cfg.py 4 5 2>&1 | grep cfg
echo "== DTQN_ROWS_FFN=32 DTQN_ROWS_WIDE=32"; DTQN_ROWS_FFN=32 DTQN_ROWS_WIDE=32 python tests/perf/time_stages_cfg.py 4 2>&1 | grep cfg
