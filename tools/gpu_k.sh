#!/bin/bash
set -u
mkdir -p gpurun_out/k
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/k/pytest.log 2>&1
tail -4 gpurun_out/k/pytest.log
timeout 900 python bench.py > gpurun_out/k/bench_default.json 2> gpurun_out/k/bench_default.err
tail -2 gpurun_out/k/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/k/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"], 1), d["update_latency_us"], {k: round(v, 1) for k, v in d["kernels_us"].items()})
print("env", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["env_steps_per_sec"].items() if k != "note"})
print("env3", d.get("env_steps_per_sec_config3"))
PY
