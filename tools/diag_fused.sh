cd "$GRAFT_REPO_ROOT"
V=$PWD/tools/variants
one() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 2000 --warmup 300 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab_$tag.json') if l.startswith('{')][0])
    print('$tag', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'])
except Exception as e:
    print('$tag failed', e, open('gpurun_out/ab_$tag.err').read()[-300:])
PY
}
one norole DTQN_HIP_LIB=$V/libdtqn_hip_norole.so
one plainld DTQN_HIP_LIB=$V/libdtqn_hip_plainld.so
echo "--- prof fused"; DTQN_HIP_LIB=$V/libdtqn_hip_prof.so python tests/perf/stage_profile.py 32 2>&1 | sed -n '/backward/,$p'
echo "--- prof unfused"; DTQN_WGRAD_FUSED=0 DTQN_HIP_LIB=$V/libdtqn_hip_prof.so python tests/perf/stage_profile.py 32 2>&1 | sed -n '/backward/,$p'
