"""Error paths of the TD update on the CPU kernel emulation (the -m gpu twins are in tests/test_gpu_error_paths.py):
  * non-finite gradient norm -- dtqn/agents/dtqn.py:257-261, clip_grad_norm_(error_if_nonfinite=True): the step is skipped,
    theta / Adam moments / step count untouched, the reference's own RuntimeError text (fixture G12, `nonfinite/*`: the
    reference's train() with infinite rewards);
  * device-side gradient exchange whose peer never publishes (dtqn_td_xreduce): bounded wait, status word, update skipped,
    the host raises instead of training on a stale sum."""
import ctypes
import os

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G12_loop.npz")


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


def _trained_agent(emu, n_good=2):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.random import set_global_seed
    from test_emu_agent import make_agent
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(4, env)
    agent = make_agent(emu, env, tuf=100, sampler="device", sample_seed=4)
    runpy.prepopulate(agent, 1400, [env])
    for _ in range(n_good):
        agent.train()
    agent._drain_stats(block=True)
    return agent


def run_nonfinite_case(agent, n_good=2):
    """Shared with the GPU twin: poison the replay rewards, train, and check what the reference's exception leaves behind."""
    fx = np.load(GOLDEN)
    eng = agent.engine
    snap = lambda: (eng.theta_pol.clone(), eng.theta_tgt.clone(), eng.adam_m.clone(), eng.adam_v.clone())
    before = snap()
    assert agent.num_train_steps == n_good and int(eng.step_counter[1]) == n_good
    agent.replay_buffer.dev.rewards.fill_(float("inf"))
    for _ in range(3):                      # the raise may come up to STATS_DRAIN_EVERY updates late: these must all be no-ops
        agent.train()
    with pytest.raises(RuntimeError) as ei:
        agent.td_errors.mean()              # readers of the running averages drain the statistics ring
    assert type(ei.value).__name__ == str(fx["nonfinite/type"])
    assert str(ei.value) == str(fx["nonfinite/message"])
    after = snap()
    assert bool(fx["nonfinite/params_untouched"])
    for a, b in zip(before, after):
        assert torch.equal(a, b), "a skipped update modified parameters or optimizer state"
    # dtqn.py:265-266 never ran: the step counts stand where they were; the loss of the failing update was logged (:253), its norm was not
    assert agent.num_train_steps == n_good + int(fx["nonfinite/num_train_steps"])
    assert int(eng.step_counter[1]) == n_good and int(eng.step_counter[3]) == 1
    assert len(agent.td_errors.q) == n_good + int(fx["nonfinite/td_errors_len"]) and not np.isfinite(agent.td_errors.q[-1])
    assert len(agent.grad_norms.q) == n_good
    st = eng.read_stats()
    assert st["nonfinite"] in (1.0, 3.0) and st["step"] == n_good + 1


def test_nonfinite_gradient_norm_skips_the_step_and_raises_like_the_reference(emu):
    run_nonfinite_case(_trained_agent(emu))


def run_dead_peer_case(lib, eng, rep, device):
    """dtqn_td_xreduce with world = 2 where rank 1 never publishes generation 1."""
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    n = eng.net.n_trainable
    mine, peer = torch.randn(n, device=device), torch.zeros(n, device=device)
    flags = torch.zeros(2, dtype=torch.int32, device=device)
    flags[0] = 1                                                   # this rank published generation 1; the peer's word stays at 0
    gptr = torch.tensor([mine.data_ptr(), peer.data_ptr()], dtype=torch.int64, device=device)
    fptr = torch.tensor([flags.data_ptr(), flags.data_ptr() + 4], dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    gsum = torch.zeros(n, device=device)
    os.environ["DTQN_XCH_TIMEOUT_MS"] = "150"
    try:
        import time
        t0 = time.time()
        assert lib.dtqn_td_xreduce(eng._net_ref, eng._td_ref, vp(gptr), vp(fptr), 2, 1, vp(gsum), vp(status), None, eng._stream()) == 0
        assert int(status.item()) == 1, "the bounded wait did not report the missing peer"
        assert time.time() - t0 < 20.0
    finally:
        del os.environ["DTQN_XCH_TIMEOUT_MS"]
    # the optimizer kernel refuses the stale sum
    eng.forward_backward(rep)
    eng.td.xstatus = status.data_ptr()
    before = (eng.theta_pol.clone(), eng.adam_m.clone(), eng.adam_v.clone())
    eng.clip_adam()
    st = eng.read_stats()
    assert st["nonfinite"] == 2.0 and int(eng.step_counter[1]) == 0 and int(eng.step_counter[3]) == 1
    for a, b in zip(before, (eng.theta_pol, eng.adam_m, eng.adam_v)):
        assert torch.equal(a, b)
    eng.td.xstatus = None
    eng.clip_adam()                                                # sticky: later calls skip too until the host has raised
    assert eng.read_stats()["nonfinite"] == 3.0 and int(eng.step_counter[1]) == 0


def test_exchange_with_a_dead_peer_sets_the_status_word_and_the_update_is_skipped(emu):
    from helpers import make_td_case
    from oracle import dtqn_oracle as O
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, num_layers=1, history_len=8)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=3, batch=4, T=20, n_eps=10, mask=-5)
    eng.set_indices(*host.sample_indices(4))
    run_dead_peer_case(emu, eng, rep, torch.device("cpu"))


def test_agent_raises_when_the_exchange_status_word_is_set(emu):
    """DtqnAgent._drain_stats turns stats[11] == 2 into the exchange error (the production path: P2PExchange points
    DtqnTd.xstatus at its status word)."""
    agent = _trained_agent(emu)
    eng = agent.engine
    status = torch.ones(1, dtype=torch.int32)
    eng.td.xstatus = status.data_ptr()
    theta = eng.theta_pol.clone()
    agent.train()
    with pytest.raises(RuntimeError, match="gradient exchange"):
        agent.grad_norms.mean()
    assert torch.equal(theta, eng.theta_pol) and agent.num_train_steps == 2
