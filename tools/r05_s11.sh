#!/bin/bash
# round 5, session 11: row-split policy beyond latency mode (BASELINE config 2 and the batches between 42 and 256 at config-1 shapes):
# whole-sequence workgroups (today) against two 32-row slices per sequence in the backward / in both kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/s11
B="--no-other-configs --no-env-rate --no-cpu-baseline"
run() {   # tag, batch, env...
  tag=$1; bs=$2; shift 2
  env "$@" timeout 120 python bench.py --config 2 --batch $bs --steps 400 --warmup 60 $B > gpurun_out/s11/$tag.json 2> gpurun_out/s11/$tag.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/s11/$tag.json') if l.startswith('{')][0])
    print('$tag batch $bs:', round(d['value'], 1), 'upd/s', round(d['ms_per_step'] * 1e3, 1), 'us', d.get('kernels_us'))
except Exception as e:
    print('$tag failed', e); print(open('gpurun_out/s11/$tag.err').read()[-400:])
PY
}
if [ "${PART:-1}" = 1 ]; then
for bs in 256 128 64; do
  run base_$bs $bs DTQN_NOP=1
  run bwd2_$bs $bs DTQN_ROW_SPLIT=1 DTQN_FWD_SLICES=1
  run both2_$bs $bs DTQN_ROW_SPLIT=1
done
run both2_nowl_256 256 DTQN_ROW_SPLIT=1 DTQN_WL=0
run bwd4_64 64 DTQN_ROW_SPLIT=4 DTQN_FWD_SLICES=1 DTQN_PIPELINE=0
else
run ride_64 64 DTQN_ROW_SPLIT=4
run ride_48 48 DTQN_ROW_SPLIT=4
run base_48 48 DTQN_NOP=1
run bwd4_48 48 DTQN_ROW_SPLIT=4 DTQN_FWD_SLICES=1 DTQN_PIPELINE=0
run base_96 96 DTQN_NOP=1
run bwd2_96 96 DTQN_ROW_SPLIT=1 DTQN_FWD_SLICES=1
run bwd4_96 96 DTQN_ROW_SPLIT=4 DTQN_FWD_SLICES=1 DTQN_PIPELINE=0
run bwd4_128 128 DTQN_ROW_SPLIT=4 DTQN_FWD_SLICES=1 DTQN_PIPELINE=0
run bwd4_256 256 DTQN_ROW_SPLIT=4 DTQN_FWD_SLICES=1 DTQN_PIPELINE=0
run bwd2_512 512 DTQN_ROW_SPLIT=1 DTQN_FWD_SLICES=1
run base_512 512 DTQN_NOP=1
fi
echo done
