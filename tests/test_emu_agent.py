"""Host logic of the agent / replay buffer / CLI on the CPU, with the kernels running on the
test-only HIP emulation: producer records -> device arrays, index draws, asynchronous statistics,
target sync cadence, checkpoint round trip, data-parallel equivalence (gloo, world_size 2)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import replay_oracle as RO

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


def make_agent(emu, env, batch=4, L=8, D=16, H=2, tuf=3, **kw):
    from dtqn_amd.agents.dtqn import DtqnAgent
    from dtqn_amd.networks.dtqn import DTQN
    from dtqn_amd.utils import env_processing as ep
    obs_len, mask = ep.get_env_obs_length(env), ep.get_env_obs_mask(env)
    disc = ep.is_discrete_env(env)

    def factory():
        m = DTQN(obs_len, env.action_space.n, 8, 0, D, H, 2, L, discrete=disc, vocab_sizes=mask + 1 if disc else None, _test_lib=emu)
        m._allow_cpu = True
        return m
    return DtqnAgent(factory, buffer_size=12 * ep.get_env_max_steps(env), device=torch.device("cpu"), env_obs_length=obs_len,
                     max_env_steps=ep.get_env_max_steps(env), obs_mask=mask, num_actions=env.action_space.n,
                     is_discrete_env=disc, batch_size=batch, context_len=L, history=L, target_update_frequency=tuf, **kw)


@pytest.mark.parametrize("env_id", ["DiscreteCarFlag-v0", "Memory-5-v0"])
def test_rollout_fills_device_replay_like_the_reference_buffer(emu, env_id):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.random import set_global_seed, RNG
    env = envs.make(env_id)
    set_global_seed(3, env)
    agent = make_agent(emu, env, D=32)
    shadow = RO.ReplayOracle(agent.replay_buffer.max_size * env._max_episode_steps, agent.env_obs_length, agent.obs_mask,
                             env._max_episode_steps, agent.context_len)
    # mirror every producer call into the oracle buffer
    rb = agent.replay_buffer
    orig = (rb.store_obs, rb.store, rb.flush)
    rb.store_obs = lambda o: (orig[0](o), shadow.store_obs(o))
    rb.store = lambda o, a, r, d, n=0: (orig[1](o, a, r, d, n), shadow.store(o, a, r, d, n))
    rb.flush = lambda: (orig[2](), shadow.flush())
    runpy.prepopulate(agent, 900 if env_id.startswith("Disc") else 400, [env])      # > 12 episodes: the ring wraps
    arrays = rb.export_arrays()
    assert np.array_equal(arrays["obss"], shadow.obss)
    assert np.array_equal(arrays["actions"], shadow.actions[:, :, 0])
    assert np.array_equal(arrays["rewards"], shadow.rewards[:, :, 0])
    assert np.array_equal(arrays["dones"].astype(bool), shadow.dones[:, :, 0])
    assert np.array_equal(arrays["eplens"], shadow.episode_lengths)
    assert np.array_equal(rb.dev.ep_len.numpy(), shadow.episode_lengths)
    assert list(rb.pos) == list(shadow.pos)
    # index draws consume Python's `random` exactly like the reference's sample()
    random.seed(5)
    e1, s1 = rb.sample_indices(6)
    random.seed(5)
    e2, s2 = shadow.sample_indices(6)
    assert np.array_equal(e1, e2) and np.array_equal(s1, s2)
    random.seed(9)
    got = rb.sample(5)
    random.seed(9)
    ref = shadow.sample(5)
    for g, r in zip(got, ref):
        assert np.array_equal(np.asarray(g).squeeze(), np.asarray(r).squeeze())


def test_train_loop_statistics_and_target_sync(emu):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import LinearAnneal
    from dtqn_amd.utils.random import set_global_seed
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(1, env)
    agent = make_agent(emu, env, tuf=3)
    agent.train()                                   # nothing to sample yet: silently returns (dtqn.py:163-164)
    assert agent.num_train_steps == 0
    runpy.prepopulate(agent, 1200, [env])
    assert agent.replay_buffer.can_sample(agent.batch_size)
    eps = LinearAnneal(1.0, 0.1, 10)
    theta0 = agent.policy_network.flat.clone()
    assert torch.equal(agent.target_network.flat, theta0)        # DqnAgent.__init__ copies policy -> target
    agent.context_reset(env.reset())
    for i in range(7):
        if runpy.step(agent, env, eps):
            agent.replay_buffer.flush()
            agent.context_reset(env.reset())
        agent.train()
        eps.anneal()
        if agent.num_train_steps in (3, 6):
            assert torch.equal(agent.target_network.flat, agent.policy_network.flat)
        elif agent.num_train_steps > 0:
            assert not torch.equal(agent.target_network.flat, agent.policy_network.flat)
    assert agent.num_train_steps == 7 and int(agent.engine.step_counter[1]) == 7
    assert not torch.equal(agent.policy_network.flat, theta0)
    assert len(agent.td_errors.q) <= 7
    m = agent.td_errors.mean()                      # drains the asynchronous ring
    assert len(agent.td_errors.q) == 7 and np.isfinite(m) and m >= 0
    assert agent.grad_norms.mean() > 0 and agent.qvalue_max.mean() >= agent.qvalue_min.mean()
    sr, ret, length = runpy.evaluate(agent, envs.make("DiscreteCarFlag-v0") if False else env, 2)
    assert 0 <= sr <= 1 and length > 0
    # the state_dict is the reference's layout and round-trips through load_state_dict
    sd = agent.policy_network.state_dict()
    assert "transformer_layers.1.attention.in_proj_weight" in sd and "transformer_layers.0.attn_mask" in sd
    agent.target_network.load_state_dict(sd)
    assert torch.equal(agent.target_network.flat, agent.policy_network.flat)


@pytest.mark.parametrize("heads", [8, 1])
def test_agent_on_the_tiled_path(emu, monkeypatch, heads):
    """Acting and training through the row-block tiled kernels behind the agent surface: forced on a small width (8 heads), and where
    dtqn_net_init sends a shape by itself (one head of width 64 -- utils/agent_utils.py:36-58 defaults num_heads=1 -- has no
    whole-sequence instantiation)."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.random import set_global_seed
    if heads == 8:
        monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(2, env)
    agent = make_agent(emu, env, batch=2, L=8, D=64, H=heads, tuf=2)
    assert agent.policy_network.net.tiled == 1
    runpy.prepopulate(agent, 600, [env])
    theta0 = agent.policy_network.flat.clone()
    agent.context_reset(env.reset())
    for _ in range(3):
        if runpy.step(agent, env, Constant(0.3)):
            agent.replay_buffer.flush(); agent.context_reset(env.reset())
        agent.train()
    assert agent.num_train_steps == 3 and np.isfinite(agent.td_errors.mean())
    assert not torch.equal(theta0, agent.policy_network.flat)
    assert not torch.equal(agent.target_network.flat, theta0)          # tuf = 2: the target was synced after update 2


def test_overlapped_step_matches_serial_step(emu):
    """begin_action / train / finish_action (two-stream pipeline on the GPU) chooses the same actions and
    reaches the same parameters as get_action + train when no episode boundary falls inside the window
    (the only semantic difference of the pipelined loop is replay visibility at episode ends)."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.random import set_global_seed
    results = []
    for overlapped in (False, True):
        env = envs.make("DiscreteCarFlag-v0")
        set_global_seed(4, env)
        agent = make_agent(emu, env, tuf=1000)
        runpy.prepopulate(agent, 1200, [env])
        eps = Constant(0.3)
        agent.context_reset(env.reset())
        acts = []
        for i in range(6):
            if overlapped:
                pending = agent.begin_action(epsilon=eps.val)
                agent.train()
                a = agent.finish_action(pending)
            else:
                # the serial loop trains AFTER the step; shift by one so both variants see update i-1 at action i
                a = agent.get_action(epsilon=eps.val)
            obs, r, done, info = env.step(a)
            assert not done
            agent.observe(obs, a, r, done)
            if not overlapped:
                agent.train()
            acts.append(int(a))
        results.append((acts, agent.policy_network.flat.clone()))
    # serial: action_i uses theta after i updates; overlapped: action_i is chosen concurrently with update i+1 but reads
    # theta after i updates as well -> identical action sequences and, after the same number of updates, identical theta
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])


@pytest.mark.parametrize("width", [(32, 2), (48, 6)])       # (48, 6): a width-padded network -- its state_dicts travel in the reference's shapes
def test_checkpoint_round_trip(emu, tmp_path, width):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import LinearAnneal
    from dtqn_amd.utils.logging_utils import RunningAverage
    from dtqn_amd.utils.random import set_global_seed
    env = envs.make("Memory-5-v0")
    set_global_seed(2, env)
    a = make_agent(emu, env, D=width[0], H=width[1])
    runpy.prepopulate(a, 400, [env])
    for _ in range(3):
        a.train()
    eps = LinearAnneal(1.0, 0.1, 10)
    eps.anneal()
    ras = [RunningAverage(10) for _ in range(3)]
    ras[0].add(0.5)
    path = str(tmp_path / "ck")
    a.save_checkpoint(path, "wid", ras[0], ras[1], ras[2], eps)
    b = make_agent(emu, env, D=width[0], H=width[1])
    wid, s, r, l, ev = b.load_checkpoint(path)
    assert wid == "wid" and ev == eps.val and s.mean() == 0.5 and b.num_train_steps == 3
    assert torch.equal(a.policy_network.flat, b.policy_network.flat) and torch.equal(a.engine.adam_v, b.engine.adam_v)
    assert torch.equal(a.replay_buffer.dev.obs, b.replay_buffer.dev.obs)
    assert b.load_mini_checkpoint(path)["step"] == 3
    # both continue identically
    st = random.getstate()
    a.train()
    random.setstate(st)          # `random` is process-global: give b the same draw
    b.train()
    assert torch.allclose(a.policy_network.flat, b.policy_network.flat, atol=0, rtol=0)
    a.td_errors.mean(); b.td_errors.mean()            # readers drain the statistics ring (the per-update drain is lazy)
    assert a.td_errors.q[-1] == b.td_errors.q[-1]      # b's statistics ring restarted cleanly


def test_cli_surface():
    import run as runpy
    a = runpy.get_args([])
    assert a.envs == ["DiscreteCarFlag-v0"] and a.model == "DTQN" and a.num_steps == 2_000_000 and a.tuf == 10_000
    assert (a.lr, a.batch, a.buf_size, a.context, a.in_embed, a.heads, a.layers) == (3e-4, 32, 500_000, 50, 128, 8, 2)
    assert (a.gate, a.pos, a.history, a.discount, a.obs_embed, a.a_embed, a.bag_size) == ("res", "learned", 50, 0.99, 8, 0, 0)
    b = runpy.get_args("--envs Memory-5-v0 --in-embed 64 --identity --pos sin --disable-wandb --sampler device".split())
    assert b.envs == ["Memory-5-v0"] and b.identity and b.pos == "sin" and b.sampler == "device"


DP_SCRIPT = r'''
import os, sys, random
sys.path.insert(0, "@REPO@"); sys.path.insert(0, "@REPO@/tests")
import numpy as np, torch, torch.distributed as td
from dtqn_amd import _binding as B, dist as ddp
from oracle import dtqn_oracle as O
from helpers import make_td_case
on_gpu = os.environ.get("DP_DEVICE", "cpu") == "cuda"
same_dev = os.environ.get("DP_SAME_DEVICE", "0") == "1"       # both ranks on cuda:0 (one-GPU box): gloo as the control plane
rank, world, local = ddp.init_from_env("cuda" if on_gpu and not same_dev else "cpu")
if on_gpu:
    from dtqn_amd import engine
    if same_dev:
        local = 0
    torch.cuda.set_device(local)
    lib, dev, kw = engine.get_lib(), f"cuda:{local}", dict(device=f"cuda:{local}", test_lib=False)
else:
    from emu import emu_build
    lib, dev, kw = B.load_library(emu_build.build()), "cpu", {}
cfg = O.NetCfg(**eval(os.environ.get("DP_CFG", "dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8)")))
Bl, T = int(os.environ.get("DP_BATCH", "4")), int(os.environ.get("DP_T", "12"))
# every rank builds the same replay; rank r trains on its own slice of a world*Bl batch
net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=Bl, T=T, n_eps=9 + 2 * Bl, mask=-5, **kw)
random.seed(77)
eps, starts = host.sample_indices(world * Bl)
if os.environ.get("DP_BREAK_P2P") == "1":
    # a device-side exchange that returns wrong sums on rank 1: the start-up check must notice and send EVERY rank to the collective
    _orig_reduce = ddp.P2PExchange.reduce
    def _bad_reduce(self):
        _orig_reduce(self)
        if self.rank == 1:
            self.engine.grad[5] += 1.0
    ddp.P2PExchange.reduce = _bad_reduce
dp = ddp.DataParallel(eng, exchange=os.environ.get("DP_EXCHANGE", "rccl"))
if os.environ.get("DP_BREAK_P2P") == "1":
    ddp.P2PExchange.reduce = _orig_reduce
with open(os.environ["OUT"] + f".sel{rank}.json", "w") as f:
    import json
    json.dump(dp.selection, f)
vote_dev = "cpu" if same_dev else dev
if not same_dev:
    dp.broadcast_parameters()           # (same-device mode: gloo cannot move device tensors; both ranks built identical parameters)
# collective votes used by DtqnAgent.train() / run.py --time-limit: every rank gets the same answer
assert ddp.agree_all(True, vote_dev) is True and ddp.agree_all(rank != 1, vote_dev) is False
assert ddp.agree_any(False, vote_dev) is False and ddp.agree_any(rank == 1, vote_dev) is True
n_updates = int(os.environ.get("DP_UPDATES", "2"))
for it in range(n_updates):
    eng.set_indices(eps[rank * Bl:(rank + 1) * Bl], starts[rank * Bl:(rank + 1) * Bl])
    dp.update(rep)
if on_gpu:
    torch.cuda.synchronize()
if dp.p2p is not None:
    dp.p2p.check()
    assert dp.p2p.k == n_updates + 2 and "p2p" in dp.exchange_kind()      # + the two generations of the start-up check
else:
    assert not eng.td.xstatus and eng.td.grad == eng.grad.data_ptr() and eng.td.xch_timeout_ms == 0   # nothing of a dropped exchange is left
np.save(os.environ["OUT"] + f".rank{rank}.npy", eng.theta_pol.cpu().numpy())
if rank == 0:
    # single learner on the union batch
    net1, _, host1, eng1, rep1 = make_td_case(lib, cfg, seed=21, batch=world * Bl, T=T, n_eps=9 + 2 * Bl, mask=-5, **kw)
    for it in range(n_updates):
        eng1.set_indices(eps, starts)
        eng1.update(rep1)
    np.save(os.environ["OUT"] + ".single.npy", eng1.theta_pol.cpu().numpy())
    np.save(os.environ["OUT"] + ".stats.npy", np.array([eng.read_stats()["grad_norm"], eng1.read_stats()["grad_norm"]]))
td.barrier()
td.destroy_process_group()
'''


def run_dp_script(tmp_path, env_extra, port, world=2):
    script = tmp_path / "dp.py"
    script.write_text(DP_SCRIPT.replace("@REPO@", REPO))
    out = str(tmp_path / "out")
    env = {**os.environ, "OUT": out, "HIPEMU_THREADS": "2", "MASTER_ADDR": "127.0.0.1", **env_extra}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0, r1, single = (np.load(out + s) for s in (".rank0.npy", ".rank1.npy", ".single.npy"))
    for r in range(1, world):
        assert np.array_equal(r0, np.load(out + f".rank{r}.npy"))          # replicas stay bit-identical
    norms = np.load(out + ".stats.npy")
    n_updates = int(env_extra.get("DP_UPDATES", "2"))
    # gradient norm of the LAST update: the trajectories of the two computations drift apart by noise-floor Adam steps (below)
    assert abs(norms[0] - norms[1]) <= (1e-5 if n_updates <= 2 else 1e-3) * norms[1]
    # Adam steps from the same state; gradients equal up to summation order (a noise-floor gradient may flip a step's sign)
    assert np.abs(r0 - single).max() <= 2.01 * n_updates * 3e-4
    assert np.mean(np.abs(r0 - single) <= 2e-6 * n_updates / 2) > (0.98 if n_updates <= 2 else 0.9)
    return r0


def test_data_parallel_equals_single_learner_gloo(emu, tmp_path):
    """world_size 2 on CPU (gloo): all-reduced half-batches == one learner on the union batch, over two updates, plus the
    collective votes DtqnAgent.train() and run.py use to keep ranks in the same control flow."""
    run_dp_script(tmp_path, {}, 29611)


def test_device_side_exchange_equals_the_all_reduce(emu, tmp_path):
    """DTQN_DP_EXCHANGE=p2p on the emulation: two processes, exchange buffers in shared memory, dtqn_xch_publish +
    dtqn_td_xreduce (flags, two generations, rank-ordered sum) instead of the collective.  Replicas bit-identical, equal to one
    learner on the union batch, and BIT-EQUAL to what the all-reduce path leaves after the same five updates (two ranks: a + b is
    the same sum in either order)."""
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    p2p = run_dp_script(tmp_path / "a", {"DP_EXCHANGE": "p2p", "DP_UPDATES": "5"}, 29613)
    ref = run_dp_script(tmp_path / "b", {"DP_EXCHANGE": "rccl", "DP_UPDATES": "5"}, 29615)
    assert np.array_equal(p2p, ref)


def test_exchange_is_selected_and_validated_at_start_up(emu, tmp_path):
    """DTQN_DP_EXCHANGE unset (auto): the device-side exchange is mapped, checked against all_reduce on known vectors for both buffer
    generations and selected; the updates that follow equal the all-reduce path's bit for bit.  With an exchange that returns a
    wrong sum on ONE rank, every rank falls back to the collective and says why."""
    import json
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir(); (tmp_path / "c").mkdir()
    auto = run_dp_script(tmp_path / "a", {"DP_EXCHANGE": "auto", "DP_UPDATES": "3"}, 29617)
    sel = [json.load(open(str(tmp_path / "a" / "out") + f".sel{r}.json")) for r in (0, 1)]
    assert all(s["kind"] == "p2p" and s["validated"] for s in sel), sel
    ref = run_dp_script(tmp_path / "b", {"DP_EXCHANGE": "rccl", "DP_UPDATES": "3"}, 29619)
    assert np.array_equal(auto, ref)
    broken = run_dp_script(tmp_path / "c", {"DP_EXCHANGE": "auto", "DP_UPDATES": "3", "DP_BREAK_P2P": "1"}, 29621)
    sel = [json.load(open(str(tmp_path / "c" / "out") + f".sel{r}.json")) for r in (0, 1)]
    assert all(s["kind"] == "rccl" and not s["validated"] and "start-up check" in s["reason"] for s in sel), sel
    assert np.array_equal(broken, ref)


@pytest.mark.parametrize("inject,needle", [("mapping:1", "peer mapping failed"), ("local:0", "start-up check"), ("sum:0", "start-up check")])
def test_injected_exchange_failure_lands_on_the_collective_everywhere(emu, tmp_path, inject, needle):
    """DTQN_DP_INJECT (dist._inject): a peer mapping that fails on ONE rank, local check work that raises on ONE rank, one wrong sum on ONE
    rank -- every rank walks through the same collectives of the start-up check, both land on the all_reduce with the reason recorded, the
    engine's exchange pointers are reset, training continues (replicas identical, equal to one learner on the union batch)."""
    import json
    run_dp_script(tmp_path, {"DP_EXCHANGE": "auto", "DP_UPDATES": "3", "DTQN_DP_INJECT": inject}, 29623 + 2 * ["mapping:1", "local:0", "sum:0"].index(inject))
    sel = [json.load(open(str(tmp_path / "out") + f".sel{r}.json")) for r in (0, 1)]
    assert all(s["kind"] == "rccl" and not s["validated"] and needle in s["reason"] for s in sel), sel


def test_vector_actor_matches_single_actor_and_reference_buffer(emu):
    """N environments, one batched actor launch per vector step (ragged prefixes): the Q rows equal what the single-actor
    entry point gives for each context alone, actions follow epsilon-greedy on them, and the episodes the actors finish are
    in the replay exactly as the reference's buffer would hold them (store_obs / store / flush per finished episode)."""
    import ctypes
    from dtqn_amd import envs
    from dtqn_amd.agents.vector import VectorActor
    from dtqn_amd.utils.random import set_global_seed, RNG
    N = 3
    env_list = [envs.make("DiscreteCarFlag-v0") for _ in range(N)]
    set_global_seed(5, *env_list)
    agent = make_agent(emu, env_list[0], batch=4, L=8, D=32, H=2)
    rb = agent.replay_buffer
    shadow = RO.ReplayOracle(rb.max_size * env_list[0]._max_episode_steps, agent.env_obs_length, agent.obs_mask,
                             env_list[0]._max_episode_steps, agent.context_len)
    orig = (rb.store_obs, rb.store, rb.flush)
    rb.store_obs = lambda o: (orig[0](o), shadow.store_obs(o))
    rb.store = lambda o, a, r, d, n=0: (orig[1](o, a, r, d, n), shadow.store(o, a, r, d, n))
    rb.flush = lambda: (orig[2](), shadow.flush())
    vec = VectorActor(agent, env_list)
    vec.reset_all()
    # desynchronise the contexts: env 1 and 2 run ahead by a few random steps (different prefix lengths in one launch)
    for i, extra in ((1, 3), (2, 11)):
        for _ in range(extra):
            obs, r, done, info = env_list[i].step(int(RNG.rng.integers(agent.num_actions)))
            vec.contexts[i].add_transition(obs, 0, r, done)
            vec.episodes[i].append((np.array(obs, copy=True), 0, float(r), bool(done)))
            assert not done
    eng = agent.engine
    for step in range(40):
        q = vec.q_values().copy()
        # the same contexts, one at a time, through the single-actor entry point
        for i, ctx in enumerate(vec.contexts):
            agent.train_context = ctx
            agent._launch_actor_forward(eng._stream())
            assert np.array_equal(agent._q_np, q[i]), (step, i)
        finished = vec.step_all(0.2)
        if finished:
            arrays = rb.export_arrays()
            assert np.array_equal(arrays["obss"], shadow.obss) and np.array_equal(arrays["rewards"], shadow.rewards[:, :, 0])
            assert np.array_equal(arrays["actions"], shadow.actions[:, :, 0]) and np.array_equal(arrays["eplens"], shadow.episode_lengths)
    assert vec.steps == 40 * N


@pytest.mark.parametrize("tiled", [False, True])
def test_agent_with_dropout_acts_in_train_mode_and_evaluates_without(emu, tiled, monkeypatch):
    """(whole-sequence kernels, and the row-block kernels forced on the same small shape)
    --dropout p: the policy network stays in train mode during rollouts (dqn.py:102-115), so two action forwards of the
    same context differ (fresh keep masks per call) and equal the oracle's forward with the same counter-based masks;
    eval_on() (run.py:206) turns dropout off: repeated forwards are identical and equal the oracle without dropout."""
    from dtqn_amd import envs
    from dtqn_amd.agents.dtqn import DtqnAgent
    from dtqn_amd.networks.dtqn import DTQN
    from dtqn_amd.utils import env_processing as ep
    from dtqn_amd.utils.random import set_global_seed
    from oracle import dtqn_oracle as O
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(2, env)
    L, D, H, p = (20, 64, 4, 0.3) if tiled else (20, 32, 4, 0.3)
    if tiled:
        monkeypatch.setenv("DTQN_FORCE_TILED", "1")

    def factory():
        m = DTQN(3, 3, 8, 0, D, H, 2, L, dropout=p, _test_lib=emu)
        m._allow_cpu = True
        return m
    agent = DtqnAgent(factory, buffer_size=12 * 200, device=torch.device("cpu"), env_obs_length=3, max_env_steps=200, obs_mask=ep.get_env_obs_mask(env),
                      num_actions=3, is_discrete_env=False, batch_size=4, context_len=L, history=L, target_update_frequency=100)
    assert agent.policy_network.net.dropout == pytest.approx(p) and agent.policy_network.net.tiled == int(tiled)
    agent.context_reset(env.reset())
    for _ in range(5):
        obs, r, done, info = env.step(1)
        agent.observe(obs, 1, r, done)
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=D, num_heads=H, num_layers=2, history_len=L, dropout=p)
    params = {k: v.detach().clone() for k, v in agent.policy_network.state_dict().items()}
    ctx = agent.context
    n = ctx.timestep + 1
    obs_t = torch.as_tensor(ctx.obs[:n], dtype=torch.float32)[None]
    act_t = torch.as_tensor(ctx.action[:n], dtype=torch.long)[None]
    eng = agent.engine
    qs = []
    for call in (1, 2):
        agent._launch_actor_forward(eng._stream())
        qs.append(agent._q_np.copy())
        spec = O.DropSpec(p, int(eng.td.dropout_seed) ^ 0xAC70, agent._actor_calls, 0)
        with torch.no_grad():
            ref = O.forward(params, cfg, obs_t, act_t, None, spec).numpy()[0, -1]
        assert np.abs(qs[-1] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (call, qs[-1], ref)
    assert not np.array_equal(qs[0], qs[1])
    agent.eval_on()
    agent.context_reset(ctx.obs[0])
    for t in range(1, n):
        agent.eval_context.add_transition(ctx.obs[t], int(ctx.action[t, 0]), 0.0, False)
    agent.eval_context.action[:n] = ctx.action[:n]
    ev = []
    for _ in range(2):
        agent._launch_actor_forward(eng._stream())
        ev.append(agent._q_np.copy())
    with torch.no_grad():
        ref = O.forward(params, cfg, obs_t, act_t).numpy()[0, -1]
    assert np.array_equal(ev[0], ev[1]) and np.abs(ev[0] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    agent.eval_off()
    assert DTQN(3, 3, 8, 0, 128, 8, 2, 128, dropout=0.1, _test_lib=emu).net.tiled == 1       # long contexts: row-block kernels, with dropout
    assert DTQN(3, 3, 8, 0, 64, 8, 2, 20, dropout=0.1, bag_size=4, _test_lib=emu).net.tiled == 1
    with pytest.raises(ValueError):
        DTQN(3, 3, 8, 0, 32, 4, 2, 20, dropout=1.5, _test_lib=emu)


def test_time_limit_vote_happens_at_vector_step_boundaries(emu, monkeypatch):
    """run.py --time-limit under data parallel with N = 8 environments: the vote (dist.agree_any) must happen once per
    TIME_CHECK_PERIOD at a vector-step boundary -- those sit at timestep = 7 (mod 8) and never at a multiple of 256 -- and
    the rank must stop and checkpoint on the iteration where the vote says so."""
    import run as runpy
    from dtqn_amd import dist as ddp, envs
    from dtqn_amd.agents.vector import VectorActor
    from dtqn_amd.utils.epsilon_anneal import LinearAnneal
    from dtqn_amd.utils.logging_utils import RunningAverage
    from dtqn_amd.utils.random import set_global_seed
    N = 8
    env_list = [envs.make("DiscreteCarFlag-v0") for _ in range(N)]
    set_global_seed(6, *env_list)
    agent = make_agent(emu, env_list[0], batch=4, L=8, D=16, H=2)
    runpy.prepopulate(agent, 1300, [env_list[0]])
    votes, saved = [], []
    monkeypatch.setattr(ddp, "is_distributed", lambda: True)
    monkeypatch.setattr(ddp, "agree_any", lambda flag, device: (votes.append(len(votes)), len(votes) >= 2)[1])
    monkeypatch.setattr(agent, "save_checkpoint", lambda *a, **k: saved.append(agent.num_train_steps))
    eps = LinearAnneal(1.0, 0.1, 100)
    monkeypatch.setattr(runpy, "time", lambda: 0.0)

    class Logger:
        def log(self, *a, **k):
            pass
    vec = VectorActor(agent, env_list)
    eval_env = envs.make("DiscreteCarFlag-v0")
    eval_env.seed(6)
    runpy.train(agent, [env_list[0]], [eval_env], ["DiscreteCarFlag-v0"], 600, eps, 10_000, 1, "/nonexistent", False,
                Logger(), RunningAverage(10), RunningAverage(10), RunningAverage(10), 1e9, False, True, False, vec)
    # votes at the first boundary of period 0 (timestep 7) and of period 1 (timestep 263); the second one stops the loop
    assert len(votes) == 2
    assert saved == [264] and agent.num_train_steps == 264


@pytest.mark.parametrize("mode,shape", [("serial", (50, 8)), ("overlap", (50, 8)), ("serial", (8, 4))])
def test_pipelined_update_equals_inline_target_pass_in_a_live_loop(emu, monkeypatch, mode, shape):
    """The pipelined update of latency mode (learner.py: next update's target pass inside the backward launch, policy passes as four
    16-row slices) on the emulation, cfg-1 network at batch 2: a pass computed ahead is only used while what it read is what the
    update would read; episode ends (replay commit, new sampling range) and hard target syncs must send the update to the inline
    pass, and the parameters must be BIT-equal to the same kernels with the target pass always inline (DTQN_PIPELINE=inline)."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import LinearAnneal
    from dtqn_amd.utils.random import set_global_seed

    def go(pipe):
        monkeypatch.setenv("DTQN_PIPELINE", pipe)
        env = envs.make("DiscreteCarFlag-v0")
        set_global_seed(8, env)
        # (8, 4): a context of 8 at head_dim 16 sits on the 64-row instantiation (dtqn_limits.h) -- three of the four row slices hold
        # no valid row at all
        agent = make_agent(emu, env, batch=2, L=shape[0], D=64, H=shape[1], tuf=5, sampler="device", sample_seed=8)
        assert agent.pipelined and agent.engine.row_split == 4 and agent.policy_network.net.lp == 64
        runpy.prepopulate(agent, 700, [env])
        eps = LinearAnneal(1.0, 1.0, 10)                     # random actions: both runs walk the same trajectory
        agent.context_reset(env.reset())
        for it in range(36):
            if it == 21:            # a host-side write to the target parameters behind the engine's back: the pass computed ahead is stale
                agent.target_network.load_state_dict(agent.policy_network.state_dict())
            if mode == "overlap":
                done = runpy.step_overlapped(agent, env, eps)
            else:
                done = runpy.step(agent, env, eps)
                agent.train()
            if done or agent.context.timestep >= 11:         # short episodes: several commits inside the run
                agent.replay_buffer.flush()
                agent.context_reset(env.reset())
        agent._drain_stats(block=True)
        e = agent.engine
        return e.theta_pol.clone(), e.theta_tgt.clone(), e.adam_v.clone(), list(agent.td_errors.q), (e._pipe["used"], e._pipe["inline"])
    a, b = go("1"), go("inline")
    used, inline = a[4]
    assert used >= 15 and inline >= 8 and b[4][0] == 0, (a[4], b[4])
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert a[3] == b[3] and len(a[3]) == 36


@pytest.mark.parametrize("seed", [1, 2])
def test_pipelined_update_under_a_random_schedule_of_everything_that_can_invalidate_it(emu, monkeypatch, tmp_path, seed):
    """The pass computed ahead must never be used after anything that changes what it read.  A seeded random schedule throws every such event
    between updates -- episode commits, hard target syncs by count and by call, load_state_dict on the target and on the policy module,
    an optimizer-state reload, a checkpoint save + load into the SAME agent, a replay re-import, a change of batch statistics range --
    and the run must be BIT-equal to the same schedule with the target pass always inline (DTQN_PIPELINE=inline)."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import LinearAnneal
    from dtqn_amd.utils.logging_utils import RunningAverage
    from dtqn_amd.utils.random import set_global_seed

    def go(pipe):
        monkeypatch.setenv("DTQN_PIPELINE", pipe)
        env = envs.make("DiscreteCarFlag-v0")
        set_global_seed(30 + seed, env)
        agent = make_agent(emu, env, batch=2, L=50, D=64, H=8, tuf=7, sampler="device", sample_seed=30 + seed)
        assert agent.pipelined
        runpy.prepopulate(agent, 500, [env])
        eps = LinearAnneal(1.0, 1.0, 10)
        sched = np.random.default_rng(seed)                   # the schedule's own stream: identical in both runs
        agent.context_reset(env.reset())
        events = []
        for it in range(30):
            ev = int(sched.integers(0, 10))
            events.append(ev)
            if ev == 0:
                agent.target_network.load_state_dict(agent.policy_network.state_dict())
            elif ev == 1:
                agent.target_update()
            elif ev == 2:
                sd = {k: v.clone() for k, v in agent.policy_network.state_dict().items()}
                agent.policy_network.load_state_dict(sd)          # same values: must change nothing, whatever the engine does about it
            elif ev == 3:
                agent.optimizer.load_state_dict(agent.optimizer.state_dict())
            elif ev == 4:
                ras = [RunningAverage(10) for _ in range(3)]
                agent.save_checkpoint(str(tmp_path / f"ck{pipe}{seed}"), "w", ras[0], ras[1], ras[2], eps)
                agent.load_checkpoint(str(tmp_path / f"ck{pipe}{seed}"))
            elif ev == 5:
                agent.replay_buffer.import_arrays(agent.replay_buffer.export_arrays())
            # 6 .. 9: nothing but the loop itself (commits at episode ends, syncs every 7 updates)
            if runpy.step(agent, env, eps) or agent.context.timestep >= 9:
                agent.replay_buffer.flush()
                agent.context_reset(env.reset())
            agent.train()
        agent._drain_stats(block=True)
        e = agent.engine
        return e.theta_pol.clone(), e.theta_tgt.clone(), e.adam_m.clone(), list(agent.td_errors.q), (e._pipe["used"], e._pipe["inline"]), events
    a, b = go("1"), go("inline")
    assert a[5] == b[5] and len(set(a[5])) >= 6                      # the schedule really mixed the events
    assert a[4][0] >= 5 and b[4][0] == 0, (a[4], b[4])               # ... and passes computed ahead were used in between
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert a[3] == b[3]


def test_data_parallel_on_a_width_padded_network(emu, tmp_path):
    """Two ranks (gloo, device-side exchange in shared memory) with a network that runs zero-padded (in-embed 48, 4 heads of 12 -> 64 columns,
    heads of 16; DtqnNet.d_real): the flat gradient that travels is the padded one, its padding is exact zeros on both ranks, so the replicas
    stay bit-identical and equal one learner on the union batch like any other shape."""
    run_dp_script(tmp_path, {"DP_EXCHANGE": "p2p", "DP_UPDATES": "3",
                             "DP_CFG": "dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=4, num_layers=1, history_len=12)"}, 29619)


def test_device_side_exchange_with_four_ranks(emu, tmp_path):
    """World size 4 on the emulation (gloo control plane, exchange buffers in shared memory): every rank waits for three peers' flags and adds
    four buffers in RANK order -- the replicas must stay bit-identical (the order is the same everywhere) and equal one learner on the
    four-fold batch; the start-up check (exact sums of integer-valued vectors) must pass whatever the order."""
    import json
    run_dp_script(tmp_path, {"DP_EXCHANGE": "auto", "DP_UPDATES": "3", "DP_BATCH": "2", "HIPEMU_THREADS": "1"}, 29623, world=4)
    for r in range(4):
        sel = json.load(open(str(tmp_path / "out") + f".sel{r}.json"))
        assert sel["kind"] == "p2p" and sel["validated"] is True, sel


def test_device_side_exchange_with_eight_ranks(emu, tmp_path):
    """World size 8 -- the node BASELINE configs 4 and 5 are defined on -- on the emulation: every rank waits for seven peers' flags and adds
    eight buffers in rank order; replicas bit-identical and equal to one learner on the eight-fold batch, exchange selected and validated
    at start-up on every rank.  (One emulation thread per rank: eight processes share this container's eight cores.)"""
    import json
    run_dp_script(tmp_path, {"DP_EXCHANGE": "auto", "DP_UPDATES": "2", "DP_BATCH": "1", "HIPEMU_THREADS": "1"}, 29627, world=8)
    for r in range(8):
        sel = json.load(open(str(tmp_path / "out") + f".sel{r}.json"))
        assert sel["kind"] == "p2p" and sel["validated"] is True, sel

