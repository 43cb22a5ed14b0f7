"""-m gpu: parity cases round 1 left on the CPU emulation only, now on the device and against the oracle:

  * the <D = 128, 64-row, one workgroup per sequence> training kernels BASELINE config 3 dispatches at its real batch;
  * the replay producer (scatter kernel reading pinned staging) bit for bit against the oracle buffer;
  * dtqn_actor_forward (pinned context in, Q[:, -1] out to pinned memory; two-workgroup latency mode for n > 32);
  * the window draw inside the forward kernel == dtqn_replay_sample for the same (seed, step);
  * Q-values at init-scale weights within an ABSOLUTE 1e-4 for every BASELINE config shape.

Each test also drops its measured errors into gpurun_out/parity_report.json (copied to profiles/ per round).
"""
import ctypes
import json
import os
import random

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O
from oracle import replay_oracle as RO

from conftest import GOLDEN, REPO
from helpers import make_td_case, check_td_updates, net_from_cfg, pack_theta, ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


def report(key, value):
    """Append a measured figure to gpurun_out/parity_report.json (pytest -q drops prints)."""
    d = os.path.join(REPO, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "parity_report.json")
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    cur[key] = float(value) if isinstance(value, (np.floating, float, int)) else value
    with open(path, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)


# ---------------------------------------------------------------------------------------------------------------
# (a) BASELINE config 3's own instantiations: D = 128, 64-row tile, ONE workgroup per sequence (3*B*2 > 256 CUs),
#     split weight gradients
# ---------------------------------------------------------------------------------------------------------------
CFG3 = dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9)


def test_cfg3_whole_sequence_kernels_at_batch_128(lib, monkeypatch):
    """Memory-5 shapes at a large batch on the whole-sequence kernels (DTQN_TRAIN_TILED=0: the library's policy would train this
    shape on the row-block kernels): row_split 1, <128, 4, 16, 8> forward / backward, 64 x 64-tile weight gradients + reduce."""
    monkeypatch.setenv("DTQN_TRAIN_TILED", "0")
    cfg = O.NetCfg(**CFG3)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=5, batch=128, T=50, n_eps=200, mask=8, device="cuda", test_lib=False)
    assert eng.row_split == 1 and net.tiled == 0 and eng.net.tiled == 0
    assert lib.dtqn_td_wgrad_is_direct(ctypes.byref(net), 128) == 0
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


def test_cfg3_training_is_routed_to_the_row_block_kernels(lib):
    """What the bench times for cfg 3: D = 128 / residual gate / post-LN / 64-row contexts beyond latency mode train on the
    row-block twin of the net (dtqn_td_prefers_tiled: same theta layout, records of the tiled kernels), while acting and
    inference forwards keep the whole-sequence kernels."""
    cfg = O.NetCfg(**CFG3)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=5, batch=128, T=50, n_eps=200, mask=8, device="cuda", test_lib=False)
    assert net.tiled == 0 and eng.net.tiled == 1 and eng.actor_net.tiled == 0
    assert eng.net.n_theta == net.n_theta and B.param_table(eng.net) == B.param_table(net)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    # inference through the engine's forward entry still takes the caller's (whole-sequence) net
    obs = torch.zeros(2, cfg.history_len, cfg.obs_dim, device="cuda")
    assert eng.forward(obs, None).shape == (2, cfg.history_len, cfg.num_actions)
    # below the threshold (latency mode reaches batch 42) the same shape trains on the whole-sequence kernels
    _, _, _, eng8, _ = make_td_case(lib, cfg, seed=5, batch=8, T=50, n_eps=20, mask=8, device="cuda", test_lib=False)
    assert eng8.net.tiled == 0


@pytest.mark.parametrize("kw,batch", [
    (dict(obs_dim=3, num_actions=5, inner_embed_size=128, num_heads=8, history_len=50), 16),
    (CFG3, 8),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50), 32),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, gate="gru"), 16),
])
def test_one_workgroup_per_sequence_forced_at_small_batch(lib, kw, batch, monkeypatch):
    """DTQN_ROW_SPLIT=0: the small-batch shapes of the latency-mode tests through the one-workgroup-per-sequence
    instantiations (what every batch > 42 runs)."""
    monkeypatch.setenv("DTQN_ROW_SPLIT", "0")
    monkeypatch.setenv("DTQN_TRAIN_TILED", "0")          # (D = 128 without row slices would otherwise train on the row-block twin)
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=23, batch=batch, T=120, n_eps=40, mask=8 if cfg.discrete else -5,
                                               device="cuda", test_lib=False)
    assert eng.row_split == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


# ---------------------------------------------------------------------------------------------------------------
# (b) replay producer kernel, bit-exact
# ---------------------------------------------------------------------------------------------------------------
def _gpu_agent(env, batch=4, L=8, D=32, H=2, **kw):
    from dtqn_amd.utils.agent_utils import get_agent
    T = env._max_episode_steps
    return get_agent("DTQN", [env], 8, 0, D, 12 * T, torch.device("cuda"), 3e-4, batch, L, -1, L, 1000, 0.99, H, 2, 0.0,
                     False, "res", "learned", 0, **kw)


@pytest.mark.parametrize("env_id,steps", [("DiscreteCarFlag-v0", 2600), ("Memory-5-v0", 900)])
def test_replay_producer_kernel_is_bit_exact(lib, env_id, steps):
    """observe() / context_reset() -> pinned staging -> dtqn_replay_push scatter kernel -> HBM arrays, against the
    oracle buffer fed the same calls (dtqn/buffers/replay_buffer.py:71-135), including slot re-use after the ring wraps,
    full staging buffers (> 256 records between commits) and the arrays a TD update then reads."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.random import set_global_seed, RNG
    env = envs.make(env_id)
    set_global_seed(3, env)
    agent = _gpu_agent(env)
    rb = agent.replay_buffer
    shadow = RO.ReplayOracle(rb.max_size * env._max_episode_steps, agent.env_obs_length, agent.obs_mask,
                             env._max_episode_steps, agent.context_len)
    orig = (rb.store_obs, rb.store, rb.flush)
    rb.store_obs = lambda o: (orig[0](o), shadow.store_obs(o))
    rb.store = lambda o, a, r, d, n=0: (orig[1](o, a, r, d, n), shadow.store(o, a, r, d, n))
    rb.flush = lambda: (orig[2](), shadow.flush())
    runpy.prepopulate(agent, steps, [env])              # > 12 episodes: the ring wraps and slots are cleansed on the device
    assert rb.pos[0] > rb.max_size
    for round_ in range(2):
        arrays = rb.export_arrays()                     # commits the queued records first
        assert np.array_equal(arrays["obss"], shadow.obss)
        assert np.array_equal(arrays["actions"], shadow.actions[:, :, 0])
        assert np.array_equal(arrays["rewards"], shadow.rewards[:, :, 0])
        assert np.array_equal(arrays["dones"].astype(bool), shadow.dones[:, :, 0])
        assert np.array_equal(rb.dev.ep_len.cpu().numpy(), shadow.episode_lengths)
        assert list(rb.pos) == list(shadow.pos)
        # an episode in progress: its records stay in the pinned staging across train() (the samplers never draw from it) and
        # reach the device with the next commit -- export_arrays above, or the flush below
        agent.context_reset(env.reset())
        for _ in range(3):
            a = int(RNG.rng.integers(env.action_space.n))
            obs, r, done, info = env.step(a)
            agent.observe(obs, a, r, done)
            agent.train()
            if done:
                break
        rb.flush()
    random.seed(9)
    got = rb.sample(5)
    random.seed(9)
    ref = shadow.sample(5)
    for g, r in zip(got, ref):
        assert np.array_equal(np.asarray(g).squeeze(), np.asarray(r).squeeze())


# ---------------------------------------------------------------------------------------------------------------
# (c) dtqn_actor_forward on the device, against G4 and the oracle
# ---------------------------------------------------------------------------------------------------------------
def _actor_forward(lib, net, theta_d, obs, act, use_workspace):
    from dtqn_amd import engine
    L, Odim, A = net.ctx_len, net.obs_dim, net.num_actions
    n = obs.shape[0]
    ctx_h = torch.zeros(L * Odim * 4 + L, dtype=torch.uint8).pin_memory()
    ctx_np = ctx_h.numpy()
    ctx_np[:L * Odim * 4].view(np.float32).reshape(L, Odim)[:n] = obs
    ctx_np[L * Odim * 4:][:n] = act
    ctx_d = torch.zeros_like(ctx_h, device="cuda")
    q_d = torch.full((L, A), float("nan"), device="cuda")
    q_last = torch.full((A,), float("nan")).pin_memory()
    ws = None
    need = lib.dtqn_forward_workspace_floats(ctypes.byref(net), 1)
    if use_workspace and need > 0:
        ws = torch.zeros(need, device="cuda")
    rc = lib.dtqn_actor_forward(ctypes.byref(net), ptr(theta_d), ptr(ctx_h), ptr(ctx_d), n, ptr(q_d), ptr(q_last),
                                None if ws is None else ptr(ws), 0, 0, 0, engine.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    if ws is not None and not net.tiled:
        assert not ws[lib.dtqn_td_xch_floats(ctypes.byref(net), 1):].any()      # hand-over flags lowered again
    return q_last.numpy().copy(), q_d.cpu().numpy()


def test_golden_G4_through_actor_forward(lib):
    """The path get_action() and bench.py's actor loop take: G4's prefixes (reference outputs) at n = 1, 2, 17, 50
    and oracle prefixes at n = 31..36, 49 (both sides of the two-workgroup switch at n > 32)."""
    z = np.load(os.path.join(GOLDEN, "G4_actor_varlen.npz"))
    worst = 0.0
    for tag in ("res", "gru_a8_sin"):
        cfg = O.NetCfg(**json.loads(str(z[f"{tag}/cfg"])))
        params = O.init_params(cfg, seed=41, perturb=True)
        net = net_from_cfg(lib, cfg)
        theta_d = torch.from_numpy(pack_theta(net, params)).cuda()
        split_capable = lib.dtqn_td_row_split(ctypes.byref(net), 1) >= 2
        assert split_capable                                   # D = 64, 64-row tile: both gates are covered
        for n in (1, 2, 17, 50):
            obs, act, ref = z[f"{tag}/n{n}_obs"][0], z[f"{tag}/n{n}_act"][0].reshape(-1), z[f"{tag}/n{n}_q"][0]
            for use_ws in (True, False):
                q_last, q_all = _actor_forward(lib, net, theta_d, obs.astype(np.float32), act, use_ws)
                err = np.abs(q_last - ref[-1]).max()
                worst = max(worst, err / max(1.0, np.abs(ref).max()))
                assert err <= 1e-4, (tag, n, use_ws, err)
                assert np.array_equal(q_last, q_all[n - 1])
                assert np.abs(q_all[:n] - ref).max() <= 1e-4
        rng = np.random.default_rng(17)
        for n in (31, 32, 33, 34, 36, 49):
            obs = rng.uniform(-1, 1, size=(n, cfg.obs_dim)).astype(np.float32)
            act = rng.integers(0, cfg.num_actions, size=n)
            with torch.no_grad():
                ref = O.forward(params, cfg, torch.as_tensor(obs[None]), torch.as_tensor(act[None, :, None], dtype=torch.long)).numpy()[0]
            for use_ws in (True, False):
                q_last, q_all = _actor_forward(lib, net, theta_d, obs, act, use_ws)
                err = np.abs(q_all[:n] - ref).max()
                worst = max(worst, err / max(1.0, np.abs(ref).max()))
                assert err <= 1e-4, (tag, n, use_ws, err)
                assert np.array_equal(q_last, q_all[n - 1])
    report("actor_forward_max_rel_err", worst)


def test_actor_forward_on_the_tiled_path(lib):
    """The same entry point for a net routed to the row-block tiled kernels (cfg 4 shapes): workspace = scratch."""
    cfg = O.NetCfg(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=128, discrete=True, vocab_sizes=12)
    params = O.init_params(cfg, seed=4, perturb=True)
    net = net_from_cfg(lib, cfg)
    assert net.tiled == 1
    theta_d = torch.from_numpy(pack_theta(net, params)).cuda()
    rng = np.random.default_rng(2)
    for n in (1, 63, 65, 128):
        obs = rng.integers(0, 11, size=(n, 6)).astype(np.float32)
        act = rng.integers(0, 6, size=n)
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs[None], dtype=torch.long), torch.as_tensor(act[None, :, None], dtype=torch.long)).numpy()[0]
        q_last, q_all = _actor_forward(lib, net, theta_d, obs, act, True)
        assert np.abs(q_all[:n] - ref).max() <= 1e-4, n
        assert np.array_equal(q_last, q_all[n - 1])


# ---------------------------------------------------------------------------------------------------------------
# (d) in-forward window draw == dtqn_replay_sample
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("batch,D", [(32, 64), (64, 64), (6, 128)])
def test_in_forward_draw_equals_replay_sample(lib, batch, D):
    """DtqnTd.sample_in_kernel: every workgroup of the forward evaluates the draw of its own sequence; the (episode,
    start) pairs it leaves behind, and the windows it read, are those dtqn_replay_sample produces for (seed, step)."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=D, num_heads=8, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=8, batch=batch, T=120, n_eps=60, mask=-5, device="cuda", test_lib=False)
    n_valid, exclude = 60, 17
    for step in (0, 1, 7, 123456):
        eng.step_counter[1] = step
        eng.sample_on_device(rep, n_valid, exclude, 99)
        torch.cuda.synchronize()
        e_ref, s_ref = eng.ep_idx.clone(), eng.start.clone()
        eng.forward_backward(rep)                     # sample_in_kernel = 0: reads the pairs above
        torch.cuda.synchronize()
        q_ref, g_ref = eng.q3.clone(), eng.grad.clone()
        eng.ep_idx.fill_(-1); eng.start.fill_(-1)
        eng.sample_in_forward(n_valid, exclude, 99)
        eng.forward_backward(rep)
        torch.cuda.synchronize()
        assert torch.equal(eng.ep_idx, e_ref) and torch.equal(eng.start, s_ref), step
        assert torch.equal(eng.q3, q_ref) and torch.equal(eng.grad, g_ref), step
        e = e_ref.cpu().numpy()
        assert (e != exclude).all() and (e >= 0).all() and (e < n_valid).all()
        lens = host.episode_lengths[e]
        assert (s_ref.cpu().numpy() <= np.maximum(0, lens - 50)).all()


# ---------------------------------------------------------------------------------------------------------------
# (e) absolute 1e-4 at init-scale weights, every BASELINE config shape
# ---------------------------------------------------------------------------------------------------------------
BASELINE_SHAPES = {
    "cfg1": (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50), 32),
    "cfg3": (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9), 16),
    "cfg4": (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2, history_len=128, discrete=True, vocab_sizes=12), 4),
    "cfg5": (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, num_layers=2, history_len=256, discrete=True, vocab_sizes=22), 2),
}


@pytest.mark.parametrize("name", sorted(BASELINE_SHAPES))
def test_q_values_absolute_tolerance_at_init_scale(lib, name):
    """north_star: 'per-timestep Q-values match the reference CPU path within 1e-4 fp32'.  With init_weights-scale
    parameters (N(0, 0.02), zero biases, LN (1, 0): |Q| << 1) the bound is absolute, not scaled by |Q|max; a second set
    with the biases / LayerNorm / positions moved off their init values is held to the same absolute bound after
    scaling the weights back to std 0.02."""
    from test_gpu_forward import hip_forward
    kw, Bn = BASELINE_SHAPES[name]
    cfg = O.NetCfg(**kw)
    rng = np.random.default_rng(31)
    worst = 0.0
    for variant in ("init", "moved"):
        params = O.init_params(cfg, seed=13, perturb=(variant == "moved"))
        if variant == "moved":
            for k, v in params.items():
                if v.dim() >= 2 and not k.endswith("attn_mask") and k != "position_embedding.position_encoding":
                    v.mul_(0.1)                          # std 0.2 -> 0.02, the init_weights scale
        n = cfg.history_len
        obs = (rng.integers(0, cfg.vocab_sizes - 1, size=(Bn, n, cfg.obs_dim)) if cfg.discrete
               else rng.uniform(-1, 1, size=(Bn, n, cfg.obs_dim)).astype(np.float32))
        act = rng.integers(0, cfg.num_actions, size=(Bn, n, 1))
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs, dtype=torch.long if cfg.discrete else torch.float32),
                            torch.as_tensor(act, dtype=torch.long)).numpy()
        got = hip_forward(lib, cfg, params, obs, act)
        err = float(np.abs(got - ref).max())
        worst = max(worst, err)
        assert np.abs(ref).max() < 1.0 and err <= 1e-4, (name, variant, err, np.abs(ref).max())
    report(f"q_abs_err_init_scale_{name}", worst)
