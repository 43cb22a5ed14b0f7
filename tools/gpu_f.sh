#!/bin/bash
set -u
mkdir -p gpurun_out/f
timeout 900 python bench.py > gpurun_out/f/bench_default.json 2> gpurun_out/f/bench_default.err
tail -3 gpurun_out/f/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/f/bench_default.json").read().strip().splitlines()[-1])
print({k: (v if not isinstance(v, dict) else "...") for k, v in d.items()})
print("latency", d["update_latency_us"]); print("env", d.get("env_steps_per_sec")); print("env3", d.get("env_steps_per_sec_config3"))
print("cpu", {k: v for k, v in d["cpu_baseline"].items() if k != "reference_in_build_container"})
for k, v in d["other_configs"].items():
    print(k, round(v["td_updates_per_s"], 1), round(v["frac_of_f32_mfma_peak"], 3), v["update_latency_us"], v["cpu_baseline"]["value"])
PY
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-env-rate --no-other-configs"
for c in 2 3; do
  DTQN_WL=0 DTQN_WAVES=4 timeout 300 $B --config $c > gpurun_out/f/bench_cfg${c}_w4.json 2>gpurun_out/f/bench_cfg${c}_w4.err
  python -c "
import json
d=json.loads(open('gpurun_out/f/bench_cfg${c}_w4.json').read().strip().splitlines()[-1])
print('cfg${c} 4 waves:', round(d['value'],1), {k: round(v,1) for k,v in d['kernels_us'].items()})" || tail -3 gpurun_out/f/bench_cfg${c}_w4.err
done
bash tools/profile_round2.sh f_prof "1" > gpurun_out/f/profile.log 2>&1
tail -40 gpurun_out/f/profile.log
