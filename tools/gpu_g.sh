#!/bin/bash
set -u
mkdir -p gpurun_out/g
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/g/pytest.log 2>&1
tail -8 gpurun_out/g/pytest.log
bash tools/profile_round2.sh g_prof "1 2 3 4 5" > gpurun_out/g/profile.log 2>&1
grep -v "^W2026" gpurun_out/g/profile.log | tail -30
