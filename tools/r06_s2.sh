#!/bin/bash
# round 6, GPU session 2: token-packed virtual rows on the row-block path -- stage times of configs 3 / 4 / 5, then the row-block GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06_s2; mkdir -p $OUT
timeout 600 python tests/perf/time_stages_cfg.py 3 4 5 --out $OUT/stages_packed.json 2>&1 | grep cfg
DTQN_PACK=0 timeout 600 python tests/perf/time_stages_cfg.py 3 --out $OUT/stages_unpacked_cfg3.json 2>&1 | grep cfg
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_td.py tests/test_gpu_forward.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
