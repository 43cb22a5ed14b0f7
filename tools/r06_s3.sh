#!/bin/bash
# round 6, GPU session 3: fragment-major weight copies (dtqn_td_wpack) + 64-row rule: stage times of configs 3 / 4 / 5 with and without, then the row-block GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06_s3; mkdir -p $OUT
echo "== packed"; timeout 600 python tests/perf/time_stages_cfg.py 3 4 5 --out $OUT/stages_wpack.json 2>&1 | grep cfg
echo "== DTQN_WPACK=0"; DTQN_WPACK=0 timeout 600 python tests/perf/time_stages_cfg.py 3 4 5 --out $OUT/stages_nopack.json 2>&1 | grep cfg
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_td.py tests/test_gpu_forward.py tests/test_gpu_pipeline.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
