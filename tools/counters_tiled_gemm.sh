#!/bin/bash
# Stall attribution with a few --pmc passes (kernel trace only), summarised per kernel:
#   bash tools/counters_tiled_gemm.sh <config> "<kernel substrings>"   ->  gpurun_out/counters_cfg<N>.txt
# SQ wait / active / instruction-mix / LDS counters (per shader engine: 8 CUs x 4 SIMDs; *_CYCLES of waves in quad-cycles).
cd "$GRAFT_REPO_ROOT" || exit 1
C=${1:-4}
KERNELS=${2:-"tl_ffn tl_wide tl_dx wgrad tl_attn"}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/counters_cfg$C.txt
: > $OUT
BENCH="python bench.py --config $C --steps 60 --warmup 10 --no-cpu-baseline --no-env-rate --no-other-configs"
i=0
for P in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 240 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc_$i -- $BENCH > /tmp/pass_$i.log 2>&1
  echo "# pass $i rc=$? : $P" >> $OUT
  DB=$(find /tmp/pmc_$i -name '*results.db' | head -1)
  if [ -n "$DB" ]; then for K in $KERNELS; do python tools/pmc_summary.py "$DB" "$K" >> $OUT 2>&1; done; fi
done
tail -40 $OUT
