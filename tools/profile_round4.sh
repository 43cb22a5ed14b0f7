#!/bin/bash
# rocprofv3 passes behind profiles/r04_*: for each BASELINE config a kernel trace + stats of bench.py, then the HBM counters
# (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, MI355X_MICROARCH.md) and one pass with the matrix-core counters
# (SQ_INSTS_VALU_MFMA_MOPS_F32, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES).  Counters run with
# --kernel-trace only.  Run on the GPU box from the repo root:
# The ONLY writer of profiles/r0N*_pmc_traffic_cfg<N>.json: tools/pmc_collect.py stamps the file with the engine's source digest
# and GIT_HEAD, which bench.py reports next to roofline.traffic.  A crashed --pmc pass is repeated once.
#   GIT_HEAD=$(git rev-parse --short HEAD) bash tools/profile_round4.sh <tag> "1 2 3 4 5"   -> gpurun_out/<tag>/cfg<N>/{kernel_stats.md, bench_under_rocprof.json, pmc_traffic.json, pmc_mfma.json}
set -u
TAG=${1:-r04}
CFGS=${2:-"1 2 3 4 5"}
export TMPDIR=/tmp
for C in $CFGS; do
  OUT=gpurun_out/$TAG/cfg$C
  mkdir -p $OUT
  STEPS=400; [ $C -ge 3 ] && STEPS=120
  BENCH="python bench.py --config $C --steps $STEPS --warmup 40 --no-cpu-baseline --no-env-rate --no-other-configs"
  rocprofv3 --kernel-trace --stats -d $OUT/kt -- $BENCH > $OUT/bench_kt.log 2>&1
  grep '^{"metric"' $OUT/bench_kt.log | tail -1 > $OUT/bench_under_rocprof.json
  DB=$(find $OUT/kt -name '*results.db' | head -1)
  python tools/rocpd_summary.py "$DB" $OUT/kernel_stats.md > /dev/null
  rm -rf $OUT/kt
  for P in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    N=$(echo $P | cut -d' ' -f1)
    for TRY in 1 2; do
      rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc_$N -- $BENCH > $OUT/bench_pmc_$N.log 2>&1
      [ -n "$(find $OUT/pmc_$N -name '*results.db' 2>/dev/null | head -1)" ] && break
      echo "pmc pass $N of cfg $C produced no database (try $TRY)"; rm -rf $OUT/pmc_$N
    done
  done
  python tools/pmc_collect.py $OUT
  python tools/pmc_finish.py $OUT $OUT/pmc_mfma_finished.json
  rm -rf $OUT/pmc_*/
  head -14 $OUT/kernel_stats.md
  cat $OUT/pmc_mfma.json | head -40
done
