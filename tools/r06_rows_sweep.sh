#!/bin/bash
# round 6: rows-per-workgroup sweep of the fused row-block kernels (32 vs 64 rows) at BASELINE configs 3 / 4 / 5, one kernel family at a time
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06_rows; mkdir -p $OUT
run() { echo "== $1"; env $1 timeout 300 python tests/perf/time_stages_cfg.py 3 4 5 --out $OUT/st_$2.json 2>&1 | grep cfg; }
run "DTQN_NOP=1" default
run "DTQN_ROWS_FFN=32" ffn32
run "DTQN_ROWS_FFN=64" ffn64
run "DTQN_ROWS_WIDE=32" wide32
run "DTQN_ROWS_WIDE=64" wide64
run "DTQN_ROWS_FFNB=32" ffnb32
run "DTQN_ROWS_FFNB=64" ffnb64
run "DTQN_GEMM_ROWS=32" gemm32
