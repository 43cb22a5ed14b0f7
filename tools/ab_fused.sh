#!/bin/bash
# A/B at cfg 1: weight gradients inside the backward launch (default) vs their own launch (DTQN_WGRAD_FUSED=0), and the number of
# weight-gradient workgroups (DTQN_FUSE_ROLE_WGS)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 3000 --warmup 300 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab_$tag.json') if l.startswith('{')][0])
    print('$tag', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'], d['update_us_median'])
except Exception as e:
    print('$tag failed', e, open('gpurun_out/ab_$tag.err').read()[-500:])
PY
}
for rep in 1 2; do
  run fused DTQN_WGRAD_FUSED=1
  run unfused DTQN_WGRAD_FUSED=0
done
run role64 DTQN_WGRAD_FUSED=1 DTQN_FUSE_ROLE_WGS=64
run role96 DTQN_WGRAD_FUSED=1 DTQN_FUSE_ROLE_WGS=96
