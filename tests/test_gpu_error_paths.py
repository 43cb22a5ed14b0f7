"""-m gpu twins of tests/test_error_paths.py: the non-finite branch of the optimizer launch (dtqn/agents/dtqn.py:257-261) and the
bounded wait of the device-side gradient exchange, on the MI355X through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(n_good=2):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils import agent_utils
    from dtqn_amd.utils.random import set_global_seed
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(4, env)
    agent = agent_utils.get_agent("DTQN", [env], 8, 0, 64, 20_000, torch.device("cuda:0"), 3e-4, 32, 50, -1, 50, 100, 0.99, 8, 2, 0.0, False,
                                  "res", "learned", 0, sampler="device", sample_seed=4)
    runpy.prepopulate(agent, 9000, [env])
    for _ in range(n_good):
        agent.train()
    agent._drain_stats(block=True)
    return agent


def test_nonfinite_gradient_norm_skips_the_step_and_raises_like_the_reference_on_the_device():
    from test_error_paths import run_nonfinite_case
    run_nonfinite_case(_agent())


def test_exchange_with_a_dead_peer_on_the_device():
    from helpers import make_td_case
    from oracle import dtqn_oracle as O
    from test_error_paths import run_dead_peer_case
    from dtqn_amd import engine
    lib = engine.get_lib()
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=3, batch=32, T=120, n_eps=40, mask=-5, device="cuda:0", test_lib=False)
    eng.set_indices(*host.sample_indices(32))
    run_dead_peer_case(lib, eng, rep, torch.device("cuda:0"))
    torch.cuda.synchronize()
