cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s5
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x > gpurun_out/s5/pipe_tests.log 2>&1
tail -12 gpurun_out/s5/pipe_tests.log
for mode in 1 0; do
  for st in 20 2000; do
    DTQN_PIPELINE=$mode timeout 300 python bench.py --steps $st --warmup $((st/4)) --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s5/bench_${mode}_${st}.json 2> gpurun_out/s5/bench_${mode}_${st}.err
  done
done
python - <<PY
import json
for mode in ('0','1'):
  for st in (20,2000):
    f=f'gpurun_out/s5/bench_{mode}_{st}.json'
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(mode, st, round(d['value'],1), round(d['ms_per_step']*1e3,2), d['kernels_us'], d.get('update_us_median'), d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e:
        print(mode, st, 'failed', e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
DTQN_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s5/trace_1 -o t -- python bench.py --steps 300 --warmup 50 --no-other-configs --no-env-rate --no-cpu-baseline > gpurun_out/s5/bench_tr.json 2> gpurun_out/s5/bench_tr.err
python - <<'PY'
import csv, glob
fs = glob.glob('gpurun_out/s5/trace_1/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if 'dtqn' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seg = rows[len(rows)//2: len(rows)//2 + 14]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    nm = r['Kernel_Name'].split('(')[0].replace('void dtqn::','')[:70]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.2f} {(int(r['End_Timestamp'])-t0)/1e3:9.2f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.2f} {nm}")
PY
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/s5/all_tests.log 2>&1
echo "all tests rc=$?" >> gpurun_out/s5/all_tests.log
tail -6 gpurun_out/s5/all_tests.log
