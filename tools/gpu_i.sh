#!/bin/bash
set -u
mkdir -p gpurun_out/i
for wl in 1 f; do
DTQN_WL=$wl DTQN_HIP_LIB=$PWD/tools/variants/libdtqn_hip_prof.so timeout 200 python tests/perf/stage_profile.py 32 > gpurun_out/i/stage_profile_B32_wl$wl.log 2>&1
sed -n 18,40p gpurun_out/i/stage_profile_B32_wl$wl.log
done
