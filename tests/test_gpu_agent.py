"""-m gpu: the agent loop on the real engine: serial vs two-stream pipelined stepping, device sampler
distribution, run.py smoke."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(seed, **kw):
    from dtqn_amd import envs
    from dtqn_amd.utils.agent_utils import get_agent
    from dtqn_amd.utils.random import set_global_seed
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(seed, env)
    agent = get_agent("DTQN", [env], 8, 0, 64, 20_000, torch.device("cuda"), 3e-4, 32, 50, -1, 50, 1000, 0.99, 8, 2, 0.0,
                      False, kw.pop("gate", "res"), "learned", 0, **kw)
    return env, agent


def test_overlapped_step_matches_serial_step_on_gpu():
    """The two-stream pipeline (actor forward || TD update, optimizer kernel waits for the actor) must be
    bit-identical to the serial loop when no episode boundary is crossed: it only reorders independent work."""
    import run as runpy
    from dtqn_amd.utils.epsilon_anneal import Constant
    results = []
    for overlapped in (False, True):
        env, agent = _agent(4)
        runpy.prepopulate(agent, 9000, [env])
        eps = Constant(0.3)
        agent.context_reset(env.reset())
        acts = []
        for i in range(40):
            if overlapped:
                pending = agent.begin_action(epsilon=eps.val)
                agent.train()
                a = agent.finish_action(pending)
            else:
                a = agent.get_action(epsilon=eps.val)
            obs, r, done, info = env.step(a)
            agent.observe(obs, a, r, done)
            if not overlapped:
                agent.train()
            acts.append(int(a))
            if done:
                break
        torch.cuda.synchronize()
        results.append((acts, agent.policy_network.flat.clone(), agent.td_errors.mean()))
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2]


def test_device_sampler_distribution():
    """dtqn_replay_sample draws episodes uniformly over finished slots minus the one in progress and starts
    uniformly on {0..max(0, len-L)} (replay_buffer.py:141-158)."""
    import run as runpy
    env, agent = _agent(2, sampler="device", sample_seed=5)
    runpy.prepopulate(agent, 9000, [env])
    rb, eng = agent.replay_buffer, agent.engine
    rb.commit()
    n_valid, exclude = rb.valid_range()
    counts = np.zeros(n_valid)
    lens = rb.episode_lengths
    for it in range(300):
        eng.step_counter[1] = it
        eng.sample_on_device(rb.dev, n_valid, exclude, 5)
        e, s = eng.ep_idx.cpu().numpy(), eng.start.cpu().numpy()
        assert (e >= 0).all() and (e < n_valid).all() and (e != exclude).all()
        assert (s >= 0).all() and (s <= np.maximum(0, lens[e] - 50)).all()
        np.add.at(counts, e, 1)
    skip = exclude < n_valid                 # the in-progress slot only lies inside [0, n_valid) once the ring has wrapped
    expected = 300 * 32 / (n_valid - (1 if skip else 0))
    if skip:
        assert counts[exclude] == 0
    live = np.delete(counts, exclude) if skip else counts
    assert abs(live.mean() - expected) < 1e-9 and live.std() < 3.5 * np.sqrt(expected)


def test_gru_agent_trains():
    import run as runpy
    from dtqn_amd.utils.epsilon_anneal import Constant
    env, agent = _agent(3, gate="gru")
    runpy.prepopulate(agent, 9000, [env])
    theta0 = agent.policy_network.flat.clone()
    agent.context_reset(env.reset())
    for _ in range(5):
        if runpy.step(agent, env, Constant(0.2)):
            agent.replay_buffer.flush(); agent.context_reset(env.reset())
        agent.train()
    assert np.isfinite(agent.td_errors.mean()) and not torch.equal(theta0, agent.policy_network.flat)
    assert "transformer_layers.1.mlp_gate.u_g.weight" in agent.policy_network.state_dict()


def test_long_context_agent_runs_on_the_tiled_path():
    """context 128 / width 128 (BASELINE config-4 class): acting and training both go through the row-block tiled
    kernels behind the same agent surface."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.agent_utils import get_agent
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.random import set_global_seed
    env = envs.make("DiscreteCarFlag-v0")           # 200-step episodes: windows of 128 fit
    set_global_seed(6, env)
    agent = get_agent("DTQN", [env], 8, 0, 128, 20_000, torch.device("cuda"), 3e-4, 8, 128, -1, 128, 1000, 0.99, 8, 2, 0.0,
                      False, "res", "learned", 0)
    assert agent.policy_network.net.tiled == 1
    runpy.prepopulate(agent, 4000, [env])
    theta0 = agent.policy_network.flat.clone()
    agent.context_reset(env.reset())
    for _ in range(6):
        if runpy.step(agent, env, Constant(0.2)):
            agent.replay_buffer.flush(); agent.context_reset(env.reset())
        agent.train()
    assert agent.num_train_steps == 6
    assert np.isfinite(agent.td_errors.mean()) and np.isfinite(agent.grad_norms.mean())
    assert not torch.equal(theta0, agent.policy_network.flat)


def test_vector_actor_on_gpu():
    """N = 5 environments through dtqn_actor_forward_batch (ragged prefixes, two-workgroup latency mode once a prefix
    passes 32 rows): every Q row equals the single-actor entry point's for the same context, and training continues
    from the episodes the vector actors commit."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.agents.vector import VectorActor
    from dtqn_amd.utils.random import set_global_seed
    N = 5
    env_list = [envs.make("DiscreteCarFlag-v0") for _ in range(N)]
    set_global_seed(7, *env_list)
    _, agent = _agent(7)
    runpy.prepopulate(agent, 9000, [env_list[0]])
    vec = VectorActor(agent, env_list)
    vec.reset_all()
    eng = agent.engine
    pos0 = agent.replay_buffer.pos[0]
    STEPS = 210               # past DiscreteCarFlag's 200-step limit: every environment has finished an episode whatever the policy does
    for step in range(STEPS):
        q = vec.q_values().copy()
        if step % 7 == 0 or step in (31, 32, 33, 49, 50, 51):
            saved = agent.train_context
            for i, ctx in enumerate(vec.contexts):
                agent.train_context = ctx
                agent._launch_actor_forward(eng._stream())
                torch.cuda.synchronize()
                assert np.abs(agent._q_np - q[i]).max() <= 1e-6 * max(1.0, np.abs(q[i]).max()), (step, i, agent._q_np, q[i])
            agent.train_context = saved
        vec.step_all(0.3)
        for _ in range(N):
            agent.train()
    torch.cuda.synchronize()
    assert vec.steps == STEPS * N and agent.num_train_steps == STEPS * N
    assert agent.replay_buffer.pos[0] == pos0 + vec.episodes_done and vec.episodes_done > 0
    assert np.isfinite(agent.td_errors.mean())
