#!/bin/bash
# round 6, GPU session 1: HEAD sanity (row-block GPU tests) + stage times of BASELINE configs 3 / 4 / 5 before this round's kernel work
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06_s1; mkdir -p $OUT
timeout 600 python tests/perf/time_stages_cfg.py 3 4 5 --out $OUT/stages_head.json 2>&1 | tail -5

