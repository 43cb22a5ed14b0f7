"""-m gpu, needs >= 2 GPUs (skipped on the one-GPU test box; runs on the 8-GPU node): two ranks on RCCL take the same
updates as one learner on the union batch -- the same script the gloo test runs on the CPU emulation, on the real
engine at BASELINE config 1 shapes (latency-mode kernels, one-launch weight gradients, flat-gradient all-reduce)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_data_parallel_equals_single_learner_rccl(tmp_path):
    from test_emu_agent import run_dp_script
    run_dp_script(tmp_path, {"DP_DEVICE": "cuda", "DP_BATCH": "32", "DP_T": "120", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                             "DP_CFG": "dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)"},
                  29631)
