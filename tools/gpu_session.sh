cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s12
for c in 5 4 3; do
 for mode in none auto; do
  if [ $mode = auto ]; then export DTQN_GEMM_ROWS=auto; else unset DTQN_GEMM_ROWS; fi
  timeout 300 python bench.py --config $c --steps 200 --warmup 20 --prewarm 50 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s12/b_${c}_$mode.json 2> gpurun_out/s12/b_${c}_$mode.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/s12/b_${c}_$mode.json') if l.startswith('{')][0])
    print('cfg$c rows=$mode', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['whole_update_frac'])
except Exception as e:
    print('failed', e); print(open('gpurun_out/s12/b_${c}_$mode.err').read()[-2000:])
PY
 done
done
