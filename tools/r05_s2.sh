#!/bin/bash
# round 5, GPU session 2: attribution of the critical-path options (dtqn_device.hpp, DTQN_OPT bits): one rocprofv3 kernel trace of the
# cfg-1 bench per variant library (tools/variants/libdtqn_hip_o<mask>.so) -> per-kernel averages inside the pipeline; the -m gpu suite on
# the product build; finer stage marks (prof variant).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/s2; mkdir -p $OUT
T0=$SECONDS
B="--no-other-configs --no-cpu-baseline --no-env-rate"
trace() {   # tag, lib
  local W=$1 LIB=$2
  DTQN_HIP_LIB=$LIB timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kt_$W -- python bench.py --steps 600 --warmup 50 $B > $OUT/bench_kt_$W.log 2>&1
  local DB=$(find $OUT/kt_$W -name '*results.db' | head -1)
  python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$W.md > /dev/null 2>&1
  rm -rf $OUT/kt_$W
  python - "$W" "$OUT/kernel_stats_$W.md" "$OUT/bench_kt_$W.log" <<'PY'
import sys, re, json
tag, f, log = sys.argv[1:4]
rows = {}
for l in open(f):
    m = re.match(r"\| `([^`]*)` \| (\d+) \| [\d.]+ \| ([\d.]+) \| ([\d.]+)", l)
    if m:
        n = m.group(1)
        k = "bwd" if "backward_kernel" in n else "fwd" if "forward_kernel" in n else "wgrad" if "wgrad" in n else "adam" if "clip_adam" in n else None
        if k and k not in rows: rows[k] = (float(m.group(3)), float(m.group(4)))
try:
    d = json.loads([l for l in open(log) if l.startswith('{')][-1]); us = d['ms_per_step'] * 1e3
except Exception: us = float('nan')
s = sum(v[0] for v in rows.values())
print(f"{tag:8s} " + " ".join(f"{k} {rows.get(k, (0, 0))[0]:6.2f} (min {rows.get(k, (0, 0))[1]:5.2f})" for k in ("fwd", "bwd", "wgrad", "adam")) + f" | sum {s:6.2f} | traced step {us:6.2f} us")
PY
}
for M in ${MASKS:-0 415 1 2 4 8 16 32 64 128 256 511}; do
  if [ $M = 415 ]; then trace prod $GRAFT_REPO_ROOT/dtqn_amd/csrc/libdtqn_hip.so; else trace o$M $GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_o$M.so; fi
done
echo "traces t=$((SECONDS - T0))s"
for M in ${RATES:-0 415 511}; do
  L=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_o$M.so; [ $M = 415 ] && L=$GRAFT_REPO_ROOT/dtqn_amd/csrc/libdtqn_hip.so
  DTQN_HIP_LIB=$L timeout 90 python bench.py --steps 2000 --warmup 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('rate o$M', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'])"
done
echo "rates t=$((SECONDS - T0))s"
DTQN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libdtqn_hip_prof.so timeout 60 python tests/perf/stage_profile.py 32 > $OUT/stage_cfg1.txt 2>&1; tail -46 $OUT/stage_cfg1.txt
echo "profile t=$((SECONDS - T0))s"
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? t=$((SECONDS - T0))s"; tail -4 $OUT/tests.log | cut -c1-300; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head -20
echo "done t=$((SECONDS - T0))s"
