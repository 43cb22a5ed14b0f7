"""-m gpu: the data-parallel update on real devices -- the same script the gloo test runs on the CPU emulation, on the real engine at
BASELINE config 1 shapes (latency-mode kernels, one-launch weight gradients).  Two ranks on RCCL need two GPUs (skipped on the
one-GPU test box; runs on the 8-GPU node); the device-side exchange runs with two processes on ONE GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


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
