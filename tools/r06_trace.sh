#!/bin/bash
# round 6: kernel trace of the stage-timing tool at one config, with the environment given:  bash tools/r06_trace.sh <cfg> <tag> [ENV=VAL ...]
cd "$GRAFT_REPO_ROOT" || exit 1
CFG=$1; TAG=$2; shift 2
OUT=gpurun_out/r06_trace/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT/kt -- python tests/perf/time_stages_cfg.py $CFG --out $OUT/stages.json > $OUT/log.txt 2>&1
DB=$(find $OUT/kt -name '*results.db' | head -1)
python tools/rocpd_summary.py "$DB" $OUT/kernel_stats.md > /dev/null
for K in tl_ffn_kernel tl_wide_kernelILi128ELi64ELb0 tl_wide_kernelILi128ELi64ELb1; do echo "$K:"; python tools/rocpd_sequence.py "$DB" $K 40; done > $OUT/sequence.txt
rm -rf $OUT/kt
head -12 $OUT/kernel_stats.md; cat $OUT/sequence.txt
