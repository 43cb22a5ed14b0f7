#!/bin/bash
# end-of-round wrap-up: cfg 4 / 5 rates, the whole -m gpu suite, smoke, the two-rank loop modes on one GPU, the default bench line.
# (Then, for the counters bench.py reports: GIT_HEAD=$(git rev-parse --short HEAD) bash tools/profile_round3.sh <tag> "1" and copy
#  gpurun_out/<tag>/cfg1/* into profiles/ -- see profiles/README.md.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in 4 5; do
    timeout 200 python bench.py --config $c --steps 300 --warmup 30 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/z_cfg${c}.json 2> gpurun_out/z_cfg${c}.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/z_cfg${c}.json') if l.startswith('{')][0])
    print('cfg$c', round(d['value'],1), d['ms_per_step'])
except Exception as e:
    print('cfg$c failed', e)
PY
done
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/z_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/z_tests.log
tail -5 gpurun_out/z_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 bash tools/dp_modes_one_gpu.sh
timeout 900 python bench.py > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/z_bench.json') if l.startswith('{')][0])
print('cfg1', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
print({k: v['updates_per_s'] for k, v in d['other_configs'].items()})
PY
