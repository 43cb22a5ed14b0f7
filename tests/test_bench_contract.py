"""bench.py: the figures it prices the kernels with and the JSON line the driver parses."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_algorithmic_flops_match_the_survey():
    import bench
    want = {1: 231_168, 2: 231_168, 3: 893_440, 4: 964_096, 5: 3_807_744}           # F_tok, SURVEY.md section 8d
    for k, v in want.items():
        assert bench.f_tok(bench.CONFIGS[k]) == v, k
    c = bench.CONFIGS[1]
    assert 5 * c["B"] * c["L"] * bench.f_tok(c) == 1_849_344_000                    # 1.849 GFLOP per update at cfg 1
    assert bench.CONFIGS[1]["B"] == 32 and bench.CONFIGS[1]["L"] == 50              # BASELINE.json metric: ctx=50 b=32
    assert bench.MFMA_F32_PEAK_TFLOPS == pytest.approx(157.3) and bench.HBM_PEAK_GBS == 8000.0


def test_pmc_traffic_reads_the_committed_profile():
    import bench
    t = bench.pmc_traffic("dtqn_forward_kernel", 32, 1)
    d = json.load(open(bench._round_profiles("pmc_traffic", 1)[0]))                 # this round's newest --pmc passes of cfg 1
    key = [k for k in d if "dtqn_forward_kernel" in k][0]
    assert t == int((2 * d[key]["FETCH_SIZE"] + d[key]["WRITE_SIZE"]) * 1024)       # gfx950: FETCH_SIZE doubled, KB units
    assert bench.pmc_traffic("dtqn_forward_kernel", 7, 1) is None                   # no profile for that batch
    t2 = bench.pmc_traffic("dtqn_backward_kernel", 256, 2)                          # every BASELINE config has its own files
    d2 = json.load(open(bench._round_profiles("pmc_traffic", 2)[0]))
    key2 = [k for k in d2 if "dtqn_backward_kernel" in k][0]
    assert t2 == int((2 * d2[key2]["FETCH_SIZE"] + d2[key2]["WRITE_SIZE"]) * 1024)
    names3 = [os.path.basename(p) for p in bench._round_profiles("pmc_traffic", 3)]   # newest suffix of the round first
    assert names3 == sorted(names3, reverse=True) and names3[0] >= "r02f_pmc_traffic_cfg3.json"
    # cfg 3 trains on the row-block kernels since r02f: the whole-sequence kernel is found in the older profile of the config
    assert bench.pmc_traffic("dtqn_backward_kernel", 512, 3) is not None
    m = bench.mfma_counters(1)
    fk = [k for k in m if "dtqn_forward_kernel" in k][0]
    assert 0.0 < m[fk]["mfma_util_vs_launch"] < 1.0 and m[fk]["SQ_VALU_MFMA_BUSY_CYCLES"] > 0
    ref = bench.reference_cpu_numbers(1)
    assert ref["runs"] and {r["threads"] for r in ref["runs"]} == {1, 8} and all(r["td_updates_per_s"] > 0 for r in ref["runs"])


@pytest.mark.gpu
def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the keys the driver and the judge read."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "200", "--warmup", "20", "--no-other-configs",
                          "--no-env-rate", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["warmup"] == 20 and d["unit"] == "TD-updates/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "batch 32" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert d["value"] == pytest.approx(1000.0 / d["ms_per_step"], rel=1e-6)
    assert d["value"] > 2000            # an order of magnitude above the CPU oracle; the tuned kernels do ~8.9k
