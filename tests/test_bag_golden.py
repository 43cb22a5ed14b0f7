"""Persistent-memory bag against tests/golden/G9_bag.npz (generated from the reference by tests/golden/make_golden.py gen_G9):
the oracle on the CPU, then the engine -- on the test-only HIP emulation here, on the MI355X in test_gpu_bag.py -- behind the
reference's own surface (DTQN.forward with bag arguments, DtqnAgent.observe / get_action / train, ReplayBuffer.sample_with_bag)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["disc", "cont"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, "G9_bag.npz"), allow_pickle=False)
    cfg = O.NetCfg(**json.loads(str(z[f"{name}_cfg"])))
    meta = json.loads(str(z[f"{name}_meta"]))
    pol = O.init_params(cfg, seed=meta["seed"], perturb=True)
    tgt = O.init_params(cfg, seed=meta["seed"] + 1, perturb=True)
    cs = O.param_checksum(pol)
    assert np.isfinite(cs) and cs > 0
    assert cs == pytest.approx(float(z[f"{name}_pol_checksum"]), rel=1e-12)
    return z, cfg, meta, pol, tgt


def golden_batch(z, name, i, discrete):
    g = lambda k: z[f"{name}_td_batch{i}_{k}"]
    ot = torch.long if discrete else torch.float32
    return O.Batch(obss=torch.as_tensor(g("obss"), dtype=ot), actions=torch.as_tensor(g("actions"), dtype=torch.long),
                   rewards=torch.as_tensor(g("rewards"), dtype=torch.float32), next_obss=torch.as_tensor(g("next_obss"), dtype=ot),
                   next_actions=torch.as_tensor(g("next_actions"), dtype=torch.long), dones=torch.as_tensor(g("dones"), dtype=torch.long),
                   bag_obss=torch.as_tensor(g("bag_obss"), dtype=ot), bag_actions=torch.as_tensor(g("bag_actions"), dtype=torch.long))


# --------------------------------------------------------------------------- oracle (CPU)
@pytest.mark.parametrize("name", NAMES)
def test_oracle_bag_forward_and_gradients_match_the_reference(name):
    z, cfg, meta, pol, tgt = load_case(name)
    ot = torch.long if cfg.discrete else torch.float32
    for n in (1, cfg.history_len // 2, cfg.history_len):
        g = lambda k: z[f"{name}_fwd{n}_{k}"]
        with torch.no_grad():
            q = O.forward(pol, cfg, torch.as_tensor(g("obs"), dtype=ot), torch.as_tensor(g("act")), bag_obss=torch.as_tensor(g("bag_obs"), dtype=ot),
                          bag_actions=torch.as_tensor(g("bag_act"))).numpy()
        assert np.abs(q - g("q")).max() <= 2e-6 * max(1.0, np.abs(g("q")).max()), n
    b0 = golden_batch(z, name, 0, cfg.discrete)
    grads, _ = O.td_gradients(pol, tgt, cfg, b0, 0.99, cfg.history_len)
    keys = O.trainable_keys(cfg)
    flat = np.concatenate([grads[k].numpy().ravel() for k in keys])
    ref = z[f"{name}_td_grad0_flat"]
    assert flat.shape == ref.shape
    assert np.abs(flat - ref).max() <= 1e-4 * np.abs(ref).max()
    learner = O.OracleLearner(cfg, pol, lr=3e-4, gamma=0.99, history=cfg.history_len, tuf=10_000, target=tgt)
    stats = json.loads(str(z[f"{name}_td_stats"]))
    st = learner.update(b0)
    for k, v in stats[0].items():
        assert st[k] == pytest.approx(v, rel=2e-4, abs=2e-5), k
    got = np.concatenate([learner.pol[k].numpy().ravel() for k in keys])
    solid = np.abs(ref) >= 1e-4 * np.abs(ref).max()
    assert np.abs(got - z[f"{name}_td_post0_flat"])[solid].max() <= 5e-7


# --------------------------------------------------------------------------- engine behind the reference's surface
def make_bag_agent(lib, cfg, meta, pol, tgt, device="cpu"):
    """lib: the emulation library (CPU tests) or None (the hipcc-built engine on `device`)."""
    from dtqn_amd.agents.dtqn import DtqnAgent
    from dtqn_amd.networks.dtqn import DTQN

    def load(m, params):
        m.load_state_dict({k: (params[k] if k in params else v) for k, v in m.state_dict().items()})

    def factory():
        m = DTQN(cfg.obs_dim, cfg.num_actions, cfg.embed_per_obs_dim, cfg.action_dim, cfg.inner_embed_size, cfg.num_heads, cfg.num_layers,
                 cfg.history_len, discrete=cfg.discrete, vocab_sizes=cfg.vocab_sizes if cfg.discrete else None, bag_size=cfg.bag_size,
                 **({"_test_lib": lib} if lib is not None else {}))
        m._allow_cpu = lib is not None
        m = m.to(device)
        load(m, pol)
        return m
    agent = DtqnAgent(factory, buffer_size=(meta["n_eps"] + 2) * meta["T"], device=torch.device(device), env_obs_length=cfg.obs_dim,
                      max_env_steps=meta["T"], obs_mask=meta["mask"], num_actions=cfg.num_actions, is_discrete_env=cfg.discrete,
                      batch_size=meta["B"], context_len=cfg.history_len, history=cfg.history_len, target_update_frequency=10_000,
                      bag_size=cfg.bag_size)
    load(agent.target_network, tgt)          # the agent hard-copies policy -> target on construction (dqn.py:49)
    return agent


def check_bag_surface(lib, name, device="cpu"):
    import dtqn_amd.utils.random as rnd
    z, cfg, meta, pol, tgt = load_case(name)
    seed = meta["seed"]
    ot = torch.long if cfg.discrete else torch.float32
    # ---- DTQN.forward with a bag
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(seed))
    agent = make_bag_agent(lib, cfg, meta, pol, tgt, device)
    for n in (1, cfg.history_len // 2, cfg.history_len):
        g = lambda k: z[f"{name}_fwd{n}_{k}"]
        q = agent.policy_network(torch.as_tensor(g("obs"), dtype=ot), torch.as_tensor(g("act")), torch.as_tensor(g("bag_obs"), dtype=ot),
                                 torch.as_tensor(g("bag_act"))).cpu().numpy()
        assert np.abs(q - g("q")).max() <= 1e-4 * max(1.0, np.abs(g("q")).max()), n
    # ---- DtqnAgent.train(): same `random` stream -> the same windows AND bags as the reference's sample_with_bag
    j = 0
    while f"{name}_ep{j}_obs" in z:
        obs, act, rew = z[f"{name}_ep{j}_obs"], z[f"{name}_ep{j}_act"], z[f"{name}_ep{j}_rew"]
        agent.context_reset(obs[0])
        for t in range(len(act)):
            agent.observe(obs[t + 1], int(act[t]), float(rew[t]), t == len(act) - 1)
        agent.replay_buffer.flush()
        j += 1
    assert j == meta["n_eps"]
    agent.eval_off()
    random.seed(seed + 7)
    state = random.getstate()
    got = agent.replay_buffer.sample_with_bag(meta["B"], agent.bag)
    names9 = ["obss", "actions", "rewards", "next_obss", "next_actions", "dones", "ep_lens", "bag_obss", "bag_actions"]
    assert len(got) == 9
    for k, a in zip(names9, got):
        ref = z[f"{name}_td_batch0_{k}"]
        assert np.array_equal(np.asarray(a).reshape(ref.shape).astype(np.float64), ref.astype(np.float64)), k
    random.setstate(state)
    pre = agent.policy_network.flat.clone()
    agent.train()
    eng = agent.engine
    # the engine's own bag gather holds the reference's bags
    assert np.array_equal(eng.bag_obs.cpu().numpy().astype(np.float64), z[f"{name}_td_batch0_bag_obss"].astype(np.float64))
    assert np.array_equal(eng.bag_actions.cpu().numpy()[..., None].astype(np.int64), z[f"{name}_td_batch0_bag_actions"].astype(np.int64))
    keys = O.trainable_keys(cfg)
    tab = B.param_table(agent.policy_network.net)
    ref_grad = z[f"{name}_td_grad0_flat"]
    gflat = eng.grad.cpu().numpy()
    got_grad = np.concatenate([gflat[tab[k][0]:tab[k][0] + int(np.prod(tab[k][1]))] for k in keys])
    assert np.abs(got_grad - ref_grad).max() <= 2e-4 * np.abs(ref_grad).max()
    stats = json.loads(str(z[f"{name}_td_stats"]))
    st = eng.read_stats()
    for k, v in stats[0].items():
        assert st[k] == pytest.approx(v, rel=2e-4, abs=2e-5), k
    post = agent.policy_network.flat.cpu().numpy()
    got_post = np.concatenate([post[tab[k][0]:tab[k][0] + int(np.prod(tab[k][1]))] for k in keys])
    solid = np.abs(ref_grad) >= 1e-3 * np.abs(ref_grad).max()
    assert np.abs(got_post - z[f"{name}_td_post0_flat"])[solid].max() <= 2e-6
    agent.train()                                        # second update: the stream stays aligned with the reference's
    assert np.array_equal(eng.bag_obs.cpu().numpy().astype(np.float64), z[f"{name}_td_batch1_bag_obss"].astype(np.float64))
    assert agent.td_errors.mean() > 0
    # ---- greedy rollout: Bag.add, then the evict-by-Q-value choice, step by step
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(seed + 3))
    agent = make_bag_agent(lib, cfg, meta, pol, tgt, device)
    agent.eval_off()
    traj = z[f"{name}_act_traj"]
    agent.context_reset(traj[0])
    for t in range(len(traj) - 1):
        a = int(agent.get_action(epsilon=0.0))
        assert a == int(z[f"{name}_act_actions"][t]), t
        agent.observe(traj[t + 1], a, 0.0, False)
        assert agent.bag.pos == int(z[f"{name}_act_bag_pos"][t]), t
        assert np.array_equal(np.asarray(agent.bag.obss, dtype=np.float64), z[f"{name}_act_bag_obss"][t]), t
        assert np.array_equal(np.asarray(agent.bag.actions, dtype=np.int64), z[f"{name}_act_bag_actions"][t]), t
    assert agent.bag.is_full
    agent.context_reset(traj[0])
    assert agent.bag.pos == 0 and np.all(agent.bag.obss == meta["mask"])


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


@pytest.mark.parametrize("name", NAMES)
def test_bag_surface_on_the_emulation(emu, name):
    check_bag_surface(emu, name)


def check_device_drawn_bags(lib, device="cpu"):
    """sampler='device': dtqn_td_forward draws the windows AND their bags (dtqn_replay_gather_bag with rows = NULL): every bag
    entry is a row of the window's episode BEFORE the window, rows distinct, the action the one stored with that row; a window
    that starts before row bag_size takes all earlier rows in order, the rest padding (replay_buffer.py:221-229)."""
    import dtqn_amd.utils.random as rnd
    z, cfg, meta, pol, tgt = load_case("cont")
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(1))
    agent = make_bag_agent(lib, cfg, meta, pol, tgt, device)
    agent.sampler, agent.sample_seed = "device", 5
    j = 0
    while f"cont_ep{j}_obs" in z:
        obs, act, rew = z[f"cont_ep{j}_obs"], z[f"cont_ep{j}_act"], z[f"cont_ep{j}_rew"]
        agent.context_reset(obs[0])
        for t in range(len(act)):
            agent.observe(obs[t + 1], int(act[t]), float(rew[t]), t == len(act) - 1)
        agent.replay_buffer.flush()
        j += 1
    seen_sampled = seen_prefix = 0
    for it in range(6):
        agent.train()
        eng, arrays = agent.engine, agent.replay_buffer.export_arrays()
        eps, starts = eng.ep_idx.cpu().numpy(), eng.start.cpu().numpy()
        bo, ba = eng.bag_obs.cpu().numpy(), eng.bag_actions.cpu().numpy()
        for b in range(meta["B"]):
            ep_obs, ep_act, st = arrays["obss"][eps[b]], arrays["actions"][eps[b]], int(starts[b])
            if st < cfg.bag_size:
                seen_prefix += 1
                assert np.array_equal(bo[b, :st], ep_obs[:st]) and np.array_equal(ba[b, :st], ep_act[:st])
                assert np.all(bo[b, st:] == meta["mask"]) and np.all(ba[b, st:] == 0)
            else:
                seen_sampled += 1
                rows = [int(np.flatnonzero((ep_obs[:st] == bo[b, k]).all(axis=1))[0]) for k in range(cfg.bag_size)]
                assert len(set(rows)) == cfg.bag_size
                assert np.array_equal(ba[b], ep_act[rows])
    assert seen_sampled > 0 and seen_prefix > 0
    assert agent.td_errors.mean() > 0


def test_device_drawn_bags_on_the_emulation(emu):
    check_device_drawn_bags(emu)


def test_bag_class_matches_the_reference_semantics():
    from dtqn_amd.utils.bag import Bag
    bag = Bag(3, -5, 2, discrete=False)
    assert bag.obss.dtype == np.float32 and bag.obss.shape == (3, 2) and bag.actions.shape == (3, 1) and not bag.is_full
    for i in range(3):
        assert bag.add(np.array([0.5 + i, -0.25]), i)
    assert bag.is_full and not bag.add(np.array([9.0, 9.0]), 1)          # a full bag rejects (utils/bag.py:29-37)
    o, a = bag.export()
    assert o.shape == (3, 2) and a[:, 0].tolist() == [0, 1, 2] and o[2, 0] == 2.5
    bag.reset()
    assert bag.pos == 0 and np.all(bag.obss == -5) and np.all(bag.actions == 0)
    assert bag.export()[0].shape == (0, 2)
    q = Bag(2, -5, 2, ref_quirks=True)                                   # the reference's dtype-less np.full: int64, truncating
    q.add(np.array([0.9, -0.9]), 1)
    assert q.obss.dtype == np.int64 and q.obss[0].tolist() == [0, 0]
    assert Bag(2, 7, 1, discrete=True).obss.dtype == np.int64


class _ScriptedEnv:
    """Replays a fixed observation sequence (the golden rollout's) behind the env surface the vectorised rollout uses."""

    def __init__(self, traj):
        self.traj, self.t = traj, 0

    def reset(self):
        self.t = 0
        return self.traj[0]

    def step(self, action):
        self.t += 1
        return self.traj[self.t], 0.0, False, {}


def check_vector_bag_rollout(lib, name, device="cpu"):
    """VectorActor with a bag network: N environments, N bags, one batched forward per vector step.  Fed the golden rollout's
    observations in two environments (one of them a step behind, so the prefixes are ragged), every environment reproduces the
    reference's greedy actions and bag contents step by step (G9's rollout, generated from the reference)."""
    import dtqn_amd.utils.random as rnd
    from dtqn_amd.agents.vector import VectorActor
    z, cfg, meta, pol, tgt = load_case(name)
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(meta["seed"] + 3))
    agent = make_bag_agent(lib, cfg, meta, pol, tgt, device)
    agent.eval_off()
    traj = z[f"{name}_act_traj"]
    vec = VectorActor(agent, [_ScriptedEnv(traj), _ScriptedEnv(traj)])
    assert vec.bags is not None and len(vec.bags) == 2
    vec.reset_all()
    # the padding actions of a fresh context are random draws (utils/context.py:50): give both contexts the golden agent's
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(meta["seed"] + 3))
    ref_ctx_actions = rnd.RNG.rng.integers(cfg.num_actions, size=(cfg.history_len, 1))
    for c in vec.contexts:
        c.action[:] = ref_ctx_actions
    # environment 1 runs one step behind: advance environment 0 alone first
    q0 = vec.q_values()
    a0 = int(np.argmax(q0[0]))
    assert a0 == int(z[f"{name}_act_actions"][0])
    obs, r, d, info = vec.envs[0].step(a0)
    ev = vec.contexts[0].add_transition(obs, a0, r, d)
    assert ev[0] is None
    steps = len(traj) - 2
    for t in range(steps):
        q = vec.q_values()
        acts = np.argmax(q, axis=1)
        assert int(acts[0]) == int(z[f"{name}_act_actions"][t + 1]) and int(acts[1]) == int(z[f"{name}_act_actions"][t]), t
        for i, env in enumerate(vec.envs):
            obs, r, d, info = env.step(int(acts[i]))
            eo, ea = vec.contexts[i].add_transition(obs, int(acts[i]), r, d)
            if eo is not None:
                agent._bag_insert(vec.bags[i], vec.contexts[i], eo, ea)
        for i, tt in ((0, t + 1), (1, t)):
            assert vec.bags[i].pos == int(z[f"{name}_act_bag_pos"][tt]), (t, i)
            assert np.array_equal(np.asarray(vec.bags[i].obss, dtype=np.float64), z[f"{name}_act_bag_obss"][tt]), (t, i)
            assert np.array_equal(np.asarray(vec.bags[i].actions, dtype=np.int64), z[f"{name}_act_bag_actions"][tt]), (t, i)
    assert vec.bags[0].is_full and vec.bags[1].is_full
    # and the public loop: one vector step through step_all keeps going from here without error, bags reset with their envs
    vec._reset(1)
    assert vec.bags[1].pos == 0


@pytest.mark.parametrize("name", NAMES)
def test_vectorised_rollout_keeps_one_bag_per_environment(emu, name):
    check_vector_bag_rollout(emu, name)


def test_vectorised_rollout_with_bags_runs_the_public_loop(emu):
    """step_all() on live environments with a bag network: contexts overflow into the per-environment bags, finished episodes are
    replayed into the buffer, and train() samples bags for them."""
    import dtqn_amd.utils.random as rnd
    from dtqn_amd import envs
    from dtqn_amd.agents.vector import VectorActor
    z, cfg, meta, pol, tgt = load_case("cont")
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(9))
    meta = {**meta, "T": 200, "n_eps": 30, "B": 2}
    agent = make_bag_agent(emu, cfg, meta, pol, tgt)
    agent.eval_off()
    es = [envs.make("DiscreteCarFlag-v0") for _ in range(3)]
    for i, e in enumerate(es):
        e.seed(i)
    vec = VectorActor(agent, es)
    vec.reset_all()
    done = 0
    for _ in range(260):
        done += vec.step_all(0.3)
    assert done >= 3 and any(b.pos > 0 for b in vec.bags)
    assert agent.replay_buffer.can_sample(agent.batch_size)
    agent.train()
    assert agent.num_train_steps == 1 and agent.td_errors.mean() >= 0


def test_bag_agent_with_dropout_acts_in_train_mode(emu):
    """--bag-size N --dropout p: the action forward of a bag network runs in train mode like the reference's (dqn.py:102-115):
    fresh keep masks per call -- context tokens, attention weights of the layers and of the bag attention -- equal to the oracle's
    forward with the same counter-based masks; under eval_on() repeated forwards are identical and equal the oracle without dropout."""
    import dtqn_amd.utils.random as rnd
    z, cfg0, meta, pol, tgt = load_case("cont")
    cfg = O.NetCfg(**{**cfg0.to_json(), "dropout": 0.25})
    rnd.RNG.rng = np.random.Generator(np.random.PCG64(2))
    from dtqn_amd.agents.dtqn import DtqnAgent
    from dtqn_amd.networks.dtqn import DTQN

    def factory():
        m = DTQN(cfg.obs_dim, cfg.num_actions, cfg.embed_per_obs_dim, cfg.action_dim, cfg.inner_embed_size, cfg.num_heads, cfg.num_layers,
                 cfg.history_len, bag_size=cfg.bag_size, dropout=cfg.dropout, _test_lib=emu)
        m._allow_cpu = True
        m.load_state_dict({k: (pol[k] if k in pol else v) for k, v in m.state_dict().items()})
        return m
    agent = DtqnAgent(factory, buffer_size=600, device=torch.device("cpu"), env_obs_length=cfg.obs_dim, max_env_steps=40, obs_mask=meta["mask"],
                      num_actions=cfg.num_actions, is_discrete_env=False, batch_size=2, context_len=cfg.history_len, history=cfg.history_len,
                      bag_size=cfg.bag_size)
    agent.eval_off()
    traj = z["cont_act_traj"]
    agent.context_reset(traj[0])
    for t in range(cfg.history_len + 3):                        # a full context and a partly filled bag
        agent.observe(traj[t + 1], int(t % cfg.num_actions), 0.0, False)
    assert 0 < agent.bag.pos < agent.bag.size           # padding entries stay in the bag forward, like the reference
    ctx, bag, eng = agent.context, agent.bag, agent.engine
    ot = torch.float32
    args = (torch.as_tensor(ctx.obs[None], dtype=ot), torch.as_tensor(ctx.action[None], dtype=torch.long))
    bargs = dict(bag_obss=torch.as_tensor(bag.obss[None], dtype=ot), bag_actions=torch.as_tensor(bag.actions[None], dtype=torch.long))
    qs = []
    for _ in range(2):
        q = agent._bag_forward(ctx.obs[None], ctx.action[None], bag.obss[None], bag.actions[None]).numpy()[0]
        spec = O.DropSpec(cfg.dropout, int(eng.td.dropout_seed) ^ 0xAC70, agent._actor_calls, 0)
        with torch.no_grad():
            ref = O.forward(pol, cfg, *args, None, spec, **bargs).numpy()[0]
        assert np.abs(q - ref).max() <= 1e-4
        qs.append(q)
    assert not np.array_equal(qs[0], qs[1])
    agent.eval_on()
    ev = [agent._bag_forward(ctx.obs[None], ctx.action[None], bag.obss[None], bag.actions[None]).numpy()[0] for _ in range(2)]
    with torch.no_grad():
        ref = O.forward(pol, cfg, *args, **bargs).numpy()[0]
    assert np.array_equal(ev[0], ev[1]) and np.abs(ev[0] - ref).max() <= 1e-4
