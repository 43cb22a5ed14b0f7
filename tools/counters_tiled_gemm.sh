#!/bin/bash
# stall attribution of the tiled GEMM kernels at cfg 4: a few --pmc passes (kernel trace only), summarised per kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > gpurun_out/s/counters.txt
wc -l gpurun_out/s/counters.txt
BENCH="python bench.py --config 4 --steps 40 --warmup 10 --no-cpu-baseline --no-env-rate --no-other-configs"
i=0
for P in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 240 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc_$i -- $BENCH > gpurun_out/s/pass_$i.log 2>&1
  echo "pass $i rc=$? : $P"
  DB=$(find /tmp/pmc_$i -name '*results.db' | head -1)
  if [ -n "$DB" ]; then python tools/pmc_summary.py "$DB" "tl_linear" > gpurun_out/s/pass_${i}_linear.txt 2>&1; python tools/pmc_summary.py "$DB" "tl_dx" > gpurun_out/s/pass_${i}_dx.txt 2>&1; python tools/pmc_summary.py "$DB" "wgrad" > gpurun_out/s/pass_${i}_wgrad.txt 2>&1; fi
  grep -v "^    @" gpurun_out/s/pass_$i.log | tail -3 > gpurun_out/s/pass_$i.tail; rm gpurun_out/s/pass_$i.log
done
cat gpurun_out/s/pass_*_linear.txt
