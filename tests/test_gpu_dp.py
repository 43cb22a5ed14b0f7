"""-m gpu: the data-parallel update on real devices -- the same script the gloo test runs on the CPU emulation, on the real engine at
BASELINE config 1 shapes (latency-mode kernels, one-launch weight gradients).  Two ranks on RCCL need two GPUs (skipped on the
one-GPU test box; runs on the 8-GPU node); the device-side exchange runs with two processes on ONE GPU."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_data_parallel_equals_single_learner_rccl(tmp_path):
    from test_emu_agent import run_dp_script
    run_dp_script(tmp_path, {"DP_DEVICE": "cuda", "DP_BATCH": "32", "DP_T": "120", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                             "DP_CFG": "dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)"},
                  29631)


def test_device_side_exchange_two_processes_on_one_gpu(tmp_path):
    """The device-side gradient exchange (DTQN_DP_EXCHANGE=p2p: dtqn_xch_publish + dtqn_td_xreduce over IPC-mapped buffers) with two
    PROCESSES on device 0 -- what a one-GPU box can run of the multi-rank path (RCCL cannot put two ranks on one GPU; gloo is only the
    control plane here): BASELINE config 1 shapes, five updates, replicas bit-identical and equal to one learner on the union batch.
    A wait that runs out sets a status word instead of hanging the GPU (bounded spin)."""
    from test_emu_agent import run_dp_script
    run_dp_script(tmp_path, {"DP_DEVICE": "cuda", "DP_SAME_DEVICE": "1", "DP_EXCHANGE": "p2p", "DP_UPDATES": "5", "DP_BATCH": "32", "DP_T": "120",
                             "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                             "DP_CFG": "dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)"},
                  29641)


def test_run_py_two_ranks_on_one_gpu(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 run.py` end to end on a one-GPU box (DTQN_DIST_SAME_DEVICE=1: both ranks on cuda:0,
    gloo for the collectives): prepopulation, the collectively decided first update, the gradient exchange inside every train(),
    evaluation on every rank, logging on rank 0 only, both ranks leaving together.  A rank that enters a collective alone hangs --
    the subprocess timeout is the assertion."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DTQN_DIST_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(REPO, "run.py"), "--disable-wandb", "--in-embed", "64", "--num-steps", "2500",
           "--prepopulate", "4000", "--eval-frequency", "1000", "--eval-episodes", "2", "--sampler", "device", "--verbose"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert out.stdout.count("Training Steps: 2000") == 1          # rank 0 alone reports


def test_run_py_two_ranks_overlapped_on_one_gpu(tmp_path):
    """Two PROCESSES on one GPU, each with the pipelined update (2 B 4 = 256-workgroup forward and backward launches whose row slices
    spin on each other's hand-overs) and an actor forward on a second stream: the combination that deadlocked the GPU in round 4 until
    a sequence's row slices were kept on ONE XCD in hand-over order (dtqn_device.hpp, slice_block_map) -- with block = sequence * RS +
    slice, one process's waiting slices could fill the XCD another process's producers needed.  The subprocess timeout is the assertion."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DTQN_DIST_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29657", os.path.join(REPO, "run.py"), "--disable-wandb", "--in-embed", "64", "--num-steps", "2500",
           "--prepopulate", "4000", "--eval-frequency", "1000", "--eval-episodes", "2", "--sampler", "device", "--verbose", "--overlap"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=150, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert out.stdout.count("Training Steps: 2000") == 1


@pytest.mark.parametrize("inject,needle", [("mapping:1", "peer mapping failed"), ("sum:1", "start-up check"), ("local:0", "start-up check")])
def test_injected_exchange_failure_sends_every_rank_to_the_collective(tmp_path, inject, needle):
    """harden the FALL-BACK.  Two processes on one GPU, DTQN_DP_EXCHANGE=auto, and a fault injected on ONE rank
    (DTQN_DP_INJECT): its peer mapping fails / its device-side exchange returns one wrong sum / its local check work raises.  Both ranks
    must land on the collective TOGETHER (a rank left alone in a collective hangs: the subprocess timeout is that assertion), say why in
    `selection`, leave the engine's xstatus / grad pointers reset, and keep training: replicas bit-identical and equal to one learner on
    the union batch."""
    import json
    from test_emu_agent import run_dp_script
    run_dp_script(tmp_path, {"DP_DEVICE": "cuda", "DP_SAME_DEVICE": "1", "DP_EXCHANGE": "auto", "DP_UPDATES": "4", "DP_BATCH": "32", "DP_T": "120",
                             "HSA_ENABLE_IPC_MODE_LEGACY": "0", "DTQN_DP_INJECT": inject,
                             "DP_CFG": "dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)"},
                  29661 + 2 * ["mapping:1", "sum:1", "local:0"].index(inject))
    sel = [json.load(open(str(tmp_path / "out") + f".sel{r}.json")) for r in (0, 1)]
    assert all(s["kind"] == "rccl" and not s["validated"] and needle in s["reason"] for s in sel), sel
    assert "injected" in sel[int(inject[-1])]["reason"] or inject.startswith("sum"), sel


def test_auto_selects_the_device_side_exchange_when_nothing_is_wrong(tmp_path):
    import json
    from test_emu_agent import run_dp_script
    run_dp_script(tmp_path, {"DP_DEVICE": "cuda", "DP_SAME_DEVICE": "1", "DP_EXCHANGE": "auto", "DP_UPDATES": "3", "DP_BATCH": "32", "DP_T": "120",
                             "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                             "DP_CFG": "dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)"},
                  29669)
    sel = [json.load(open(str(tmp_path / "out") + f".sel{r}.json")) for r in (0, 1)]
    assert all(s["kind"] == "p2p" and s["validated"] for s in sel), sel
