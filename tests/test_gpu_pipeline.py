"""-m gpu: the pipelined forward of latency mode (dtqn_amd/learner.py: target pass of update k + 1 launched ahead on a side stream,
policy passes as four 16-row slices).  A pass launched ahead may only be USED if it read what the update would read; these tests
drive the cases that invalidate it (replay commits, changed sampling range, hard target syncs, checkpoint loads) and require
BIT-equal parameters against the same kernels with the target pass always inline (DTQN_PIPELINE=inline)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, monkeypatch, steps=400, tuf=7, overlap=False):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils import agent_utils, epsilon_anneal
    from dtqn_amd.utils.random import set_global_seed
    monkeypatch.setenv("DTQN_PIPELINE", mode)
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(11, env)
    agent = agent_utils.get_agent("DTQN", [env], 8, 0, 64, 40_000, torch.device("cuda:0"), 3e-4, 32, 50, -1, 50, tuf, 0.99, 8, 2, 0.0, False,
                                  "res", "learned", 0, sampler="device", sample_seed=11)
    runpy.prepopulate(agent, 9000, [env])
    eps = epsilon_anneal.LinearAnneal(1.0, 0.5, steps)
    agent.context_reset(env.reset())
    for i in range(steps):
        done = runpy.step_overlapped(agent, env, eps) if overlap else runpy.step(agent, env, eps)
        if done:
            agent.replay_buffer.flush()
            agent.context_reset(env.reset())
        if not overlap:
            agent.train()
        eps.anneal()
    agent._drain_stats(block=True)
    torch.cuda.synchronize()
    e = agent.engine
    pipe = getattr(e, "_pipe", None)
    return (e.theta_pol.clone(), e.theta_tgt.clone(), e.adam_m.clone(), e.adam_v.clone(), list(agent.td_errors.q), agent.num_train_steps,
            None if pipe is None else (pipe["used"], pipe["inline"]))


@pytest.mark.parametrize("overlap", [False, True])
def test_target_pass_launched_ahead_equals_inline_in_a_live_loop(monkeypatch, overlap):
    a = _run("1", monkeypatch, overlap=overlap)
    b = _run("inline", monkeypatch, overlap=overlap)
    assert a[6] is not None and b[6] is not None, "the pipelined forward did not engage at cfg-1 shapes"
    used, inline = a[6]
    # most updates take the pass launched ahead; episode ends (replay commit, new sampling range) and target syncs fall back
    assert used > 0.6 * a[5] and inline >= a[5] // 7 and b[6][0] == 0
    assert a[5] == b[5] == 400
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    assert a[4] == b[4]


def test_pipelined_update_matches_the_whole_launch(monkeypatch):
    """Four-slice policy passes + separate target pass against the three-pass launch of two slices: same windows, Q and loss equal
    up to fp32 summation order (the 16-row items split their accumulation chain in two)."""
    a = _run("1", monkeypatch, steps=60, tuf=1000)
    c = _run("0", monkeypatch, steps=60, tuf=1000)
    assert c[6] is None
    ea, ec = np.array(a[4]), np.array(c[4])
    assert ea.shape == ec.shape and np.abs(ea[:10] - ec[:10]).max() <= 1e-4 * max(1.0, np.abs(ec[:10]).max())


def test_row_block_net_target_pass_on_the_side_stream_equals_inline(monkeypatch):
    """BASELINE config 5 shapes (ctx 256, d_model 256, batch 32: row-block kernels, 128 workgroups per pass): the second-stream
    flavour of the pipelined update -- the next update's target pass launched on a side stream beside this update's backward kernels --
    against the same kernels with the target pass inline: bit-equal parameters and statistics."""
    import bench

    def go(mode, n=24):
        monkeypatch.setenv("DTQN_PIPELINE", mode)
        c = bench.CONFIGS[5]
        torch.manual_seed(5)                  # the synthetic env is not seeded: same initial parameters for both runs
        agent = bench.make_agent(c, c["B"], torch.device("cuda:0"), 0, "device")
        pipe = getattr(agent.engine, "_pipe", None)
        assert pipe is not None and not pipe["ride"]
        for i in range(n):
            if i == 9:                        # a replay write in between: the pass launched ahead must be dropped once
                agent.replay_buffer.version += 1
            agent.train()
        agent._drain_stats(block=True)
        torch.cuda.synchronize()
        e = agent.engine
        return e.theta_pol.clone(), e.adam_v.clone(), list(agent.td_errors.q), (pipe["used"], pipe["inline"])
    a, b = go("1"), go("inline")
    assert a[3][0] >= 20 and a[3][1] == 2 and b[3][0] == 0, (a[3], b[3])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
