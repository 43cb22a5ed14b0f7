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


def test_pmc_traffic_reads_the_committed_profile(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "build_digest", lambda: "0" * 16)                    # no engine needed for the lookup itself
    t, src = bench.pmc_traffic("dtqn_forward_kernel", 32, 1)
    path = bench._round_profiles("pmc_traffic", 1)[0]                               # the newest --pmc passes of cfg 1
    d = json.load(open(path))
    key = [k for k in d if "dtqn_forward_kernel" in k][0]
    assert t == int((2 * d[key]["FETCH_SIZE"] + d[key]["WRITE_SIZE"]) * 1024)       # gfx950: FETCH_SIZE doubled, KB units
    assert src["file"] == os.path.relpath(path, REPO) and src["matches_build"] is False   # stamped with where it came from
    assert bench.pmc_traffic("dtqn_forward_kernel", 7, 1) == (None, None)           # no profile for that batch
    t2, _ = bench.pmc_traffic("dtqn_backward_kernel", 256, 2)                       # every BASELINE config has its own files
    assert t2 is not None and t2 > 0
    names3 = [os.path.basename(p) for p in bench._round_profiles("pmc_traffic", 3)]   # newest round / suffix first
    assert names3 == sorted(names3, reverse=True) and names3[0] >= "r02f_pmc_traffic_cfg3.json"
    m = bench.mfma_counters(1)["counters"]
    fk = [k for k in m if "dtqn_forward_kernel" in k][0]
    assert 0.0 < m[fk]["mfma_util_vs_launch"] < 1.0 and m[fk]["SQ_VALU_MFMA_BUSY_CYCLES"] > 0
    ref = bench.reference_cpu_numbers(1)
    assert ref["runs"] and {r["threads"] for r in ref["runs"]} == {1, 8} and all(r["td_updates_per_s"] > 0 for r in ref["runs"])


def test_gpus_n_outside_torchrun_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2` with no WORLD_SIZE re-executes under torch.distributed.run on 127.0.0.1."""
    import argparse
    import bench
    import torch
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(os, "execvpe", lambda exe, cmd, env: seen.update(exe=exe, cmd=cmd, env=env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "7", "--warmup", "3"])
    bench._self_launch(argparse.Namespace(gpus=2))
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "2", "--steps", "7", "--warmup", "3"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench._self_launch(argparse.Namespace(gpus=2))


def test_line_rounding_keeps_the_line_small():
    import bench
    big = {"a": 1.23456789012345, "b": {"c": [0.000123456789, 3]}, "s": "x"}
    assert bench._r(big) == {"a": 1.235, "b": {"c": [0.0001235, 3]}, "s": "x"}
    assert bench.LINE_LIMIT <= 4096


@pytest.mark.gpu
def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the keys the driver and the judge read."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "200", "--warmup", "20", "--no-other-configs",
                          "--no-env-rate", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) <= 4096                        # the driver keeps an 8 KB stdout tail: the whole line must fit
    assert lines[0].startswith('{"metric":')            # contract keys lead the line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["warmup"] == 20 and d["unit"] == "TD-updates/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "batch 32" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
    assert "traffic" in r and "traffic_src" in r
    hk = d["hbm_kernels"]                               # the HBM-bound launches priced against 8 TB/s
    assert {"clip_adam", "replay_sample", "replay_apply"} <= set(hk) and all(0 < v["frac_of_8TBs"] < 1 for v in hk.values())
    det = json.load(open(os.path.join(REPO, d["detail_file"])))
    assert det["line"]["value"] == d["value"] and "update_latency_us" in det
    assert d["value"] == pytest.approx(1000.0 / d["ms_per_step"], rel=1e-6)
    assert d["value"] > 2000            # an order of magnitude above the CPU oracle; the tuned kernels do ~8.9k


@pytest.mark.gpu
def test_two_ranks_print_one_whole_job_line_on_a_one_gpu_box():
    """`python bench.py --gpus 2` launches its own two ranks; with DTQN_DIST_SAME_DEVICE=1 both sit on cuda:0 and gloo carries the
    gradient all-reduce (RCCL refuses two ranks on one GPU), which is what a one-GPU box can run of the N > 1 path: rendezvous on
    127.0.0.1, barrier + max-over-ranks timing, the whole-job value, ONE line from rank 0.  The numbers are not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DTQN_DIST_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "10",
                          "--no-other-configs", "--no-env-rate", "--no-cpu-baseline"], capture_output=True, text=True, timeout=240,
                         cwd=REPO, env=env)        # (a rank-0-only collective hangs: this is the test that found update_latency's)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 60 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 64
    assert d["value"] == pytest.approx(2 * 1000.0 / d["ms_per_step"], rel=1e-6) and d["value"] > 0
    # the exchange the start-up check selected (two processes on one GPU: the device-side exchange maps and validates), timed by itself,
    # beside the solo rate this job measured and the efficiency that follows from it
    ex = d["exchange"]
    assert ex["kind"] in ("p2p", "rccl") and isinstance(ex["validated"], bool) and ex["reason"]
    assert ex["selected_us"]["median"] > 0 and ex["solo_updates_per_s_rank0"] > 0 and ex["weak_scaling_efficiency"] > 0
    assert len(ex["kernels_us_sum_per_rank"]) == 2 and all(v > 0 for v in ex["kernels_us_sum_per_rank"])
    assert ex["kind"] == "p2p" and ex["validated"], ex      # no environment variable needed to take the fast path


@pytest.mark.gpu
def test_two_ranks_of_config_4_on_a_one_gpu_box():
    """`python bench.py --gpus 2 --config 4` -- the first command for an 8-GPU node (README.md), BASELINE's 8 x MI355X data-parallel config
    at its per-GPU batch of 128 -- dry-run with both ranks on cuda:0 (DTQN_DIST_SAME_DEVICE=1): the row-block kernels (fused layer
    launches, LDS weight gradients, split-K reduce) through the gradient exchange and the rank-synchronous clip + Adam at least once on
    a device.  The numbers are not a measurement (two processes share one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DTQN_DIST_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "10", "--warmup", "3",
                          "--no-other-configs", "--no-env-rate", "--no-cpu-baseline"], capture_output=True, text=True, timeout=400,
                         cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 256 and "gv_memory" in d["config"]["workload"]
    assert d["value"] == pytest.approx(2 * 1000.0 / d["ms_per_step"], rel=1e-6) and d["value"] > 0
    assert d["exchange"]["kind"] in ("p2p", "rccl")

