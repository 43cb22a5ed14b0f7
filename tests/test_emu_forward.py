"""Kernel-logic tests on the CPU: the HIP sources compiled against the test-only HIP emulation
(tests/emu) versus the oracle.  These do not replace the -m gpu parity tests; they validate
indexing / LDS layout / barrier structure / MFMA fragment maps without a GPU."""
import ctypes

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from helpers import net_from_cfg, pack_theta, ptr


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


def _fwd(emu, cfg, params, obs, act):
    net = net_from_cfg(emu, cfg)
    theta = pack_theta(net, params)
    Bn, n = obs.shape[:2]
    q = np.full((Bn, n, cfg.num_actions), np.nan, dtype=np.float32)
    obs_f = np.ascontiguousarray(obs, dtype=np.float32)
    act_u8 = np.ascontiguousarray(act.reshape(Bn, n), dtype=np.uint8)
    rc = emu.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs_f), ptr(act_u8), Bn, n, ptr(q), None)
    assert rc == 0
    return q


CASES = [
    dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8),
    dict(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, history_len=20, action_dim=4),
    dict(obs_dim=10, num_actions=5, inner_embed_size=32, num_heads=2, history_len=12, discrete=True, vocab_sizes=9),
    dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, identity=True, pos="sin"),
    dict(obs_dim=1, num_actions=5, inner_embed_size=32, num_heads=4, history_len=30, discrete=True, vocab_sizes=22,
         action_dim=8, pos="none"),
    dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, gate="gru"),
    dict(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, history_len=20, gate="gru", identity=True),
    # contexts shorter than the smallest instantiated row-tile count of their (d_model, head_dim): dtqn_limits.h
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=10, num_layers=1),
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, history_len=20, num_layers=1, gate="gru"),
    dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, history_len=20, num_layers=1),
]


@pytest.mark.parametrize("kw", CASES)
def test_forward_small_variants(emu, kw):
    cfg = O.NetCfg(**kw)
    params = O.init_params(cfg, seed=3, perturb=True)
    rng = np.random.default_rng(5)
    for n in sorted({1, 2, cfg.history_len // 2, cfg.history_len}):
        Bn = 3
        if cfg.discrete:
            obs = rng.integers(0, cfg.vocab_sizes, size=(Bn, n, cfg.obs_dim))
        else:
            obs = rng.uniform(-1, 1, size=(Bn, n, cfg.obs_dim)).astype(np.float32)
        act = rng.integers(0, cfg.num_actions, size=(Bn, n, 1))
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs, dtype=torch.long if cfg.discrete else torch.float32),
                            torch.as_tensor(act, dtype=torch.long)).numpy()
        got = _fwd(emu, cfg, params, obs, act)
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= 1e-4, (n, np.abs(got - ref).max())


def test_forward_cfg1_size(emu):
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    params = O.init_params(cfg, seed=11, perturb=True)
    rng = np.random.default_rng(7)
    obs = rng.uniform(-1, 1, size=(2, 50, 3)).astype(np.float32)
    act = rng.integers(0, 3, size=(2, 50, 1))
    with torch.no_grad():
        ref = O.forward(params, cfg, torch.as_tensor(obs), torch.as_tensor(act)).numpy()
    got = _fwd(emu, cfg, params, obs, act)
    assert np.abs(got - ref).max() <= 1e-4


TILED = [
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50),
    dict(obs_dim=6, num_actions=6, inner_embed_size=64, num_heads=4, history_len=70, discrete=True, vocab_sizes=12, identity=True, pos="sin"),
    dict(obs_dim=1, num_actions=5, inner_embed_size=64, num_heads=2, history_len=130, discrete=True, vocab_sizes=22, action_dim=8),
    dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=2, history_len=70, gate="gru"),
]


@pytest.mark.parametrize("kw", TILED)
def test_tiled_forward_path(emu, kw, monkeypatch):
    """Row-block tiled kernels (the path of BASELINE configs 4 / 5), forced on small widths so the CPU emulation
    finishes quickly: multi-block contexts (L > 64), variable length, identity / discrete / action variants."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    cfg = O.NetCfg(**kw)
    net = net_from_cfg(emu, cfg)
    assert net.tiled == 1 and net.lp % 64 == 0
    params = O.init_params(cfg, seed=3, perturb=True)
    theta = pack_theta(net, params)
    rng = np.random.default_rng(5)
    for n in sorted({1, cfg.history_len // 2 + 1, cfg.history_len}):
        Bn = 2
        obs = (rng.integers(0, cfg.vocab_sizes, size=(Bn, n, cfg.obs_dim)) if cfg.discrete
               else rng.uniform(-1, 1, size=(Bn, n, cfg.obs_dim)).astype(np.float32))
        act = rng.integers(0, cfg.num_actions, size=(Bn, n, 1))
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs, dtype=torch.long if cfg.discrete else torch.float32),
                            torch.as_tensor(act, dtype=torch.long)).numpy()
        q = np.full((Bn, n, cfg.num_actions), np.nan, dtype=np.float32)
        ws = np.zeros(emu.dtqn_forward_workspace_floats(ctypes.byref(net), Bn), dtype=np.float32)
        obs_f = np.ascontiguousarray(obs, dtype=np.float32)
        act_u8 = np.ascontiguousarray(act.reshape(Bn, n), dtype=np.uint8)
        rc = emu.dtqn_forward_tiled(ctypes.byref(net), ptr(theta), ptr(obs_f), ptr(act_u8), Bn, n, ptr(q), ptr(ws), None)
        assert rc == 0
        assert np.abs(q - ref).max() <= 1e-4, (n, np.abs(q - ref).max())
    # the whole-sequence kernels refuse a tiled net, and the training entry points are not built for it
    assert emu.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs_f), ptr(act_u8), Bn, n, ptr(q), None) == B.DEFINES["DTQN_ERR_CONFIG"]


def test_variants_beyond_the_lds_tile_take_the_tiled_path(emu):
    """GRU gates and identity-reordered layers at D = 128 need more LDS than one workgroup has on the whole-sequence
    kernels: dtqn_net_init routes them to the row-block tiled path (forward and training); D = 64 stays whole-sequence."""
    for kw, tiled in ((dict(inner_embed_size=128, gate="gru"), 1), (dict(inner_embed_size=128, identity=True), 1),
                      (dict(inner_embed_size=128), 0), (dict(inner_embed_size=64, gate="gru", identity=True), 0)):
        cfg = O.NetCfg(obs_dim=3, num_actions=3, num_heads=8, num_layers=1, history_len=50, **kw)
        net = net_from_cfg(emu, cfg)
        assert net.tiled == tiled and net.lp == 64, kw
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, num_layers=1, history_len=9, gate="gru")
    net = net_from_cfg(emu, cfg)
    params = O.init_params(cfg, seed=3, perturb=True)
    theta = pack_theta(net, params)
    rng = np.random.default_rng(5)
    obs = rng.uniform(-1, 1, size=(1, 9, 3)).astype(np.float32)
    act = np.zeros((1, 9), dtype=np.uint8)
    with torch.no_grad():
        ref = O.forward(params, cfg, torch.as_tensor(obs), torch.as_tensor(act[..., None], dtype=torch.long)).numpy()
    q = np.full((1, 9, 3), np.nan, dtype=np.float32)
    ws = np.zeros(emu.dtqn_forward_workspace_floats(ctypes.byref(net), 1), dtype=np.float32)
    assert emu.dtqn_forward_tiled(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 1, 9, ptr(q), ptr(ws), None) == 0
    assert np.abs(q - ref).max() <= 1e-4


@pytest.mark.parametrize("kw", [dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50),
                                dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=55, action_dim=8)])
def test_actor_forward_one_call(emu, kw, monkeypatch):
    """dtqn_actor_forward: packed pinned context -> device -> forward -> Q[:, -1] in host memory, with the two-workgroup
    latency mode (workspace given, n > 32) and without; both against the oracle."""
    monkeypatch.setenv("DTQN_ROW_SPLIT", "1")
    cfg = O.NetCfg(**kw)
    params = O.init_params(cfg, seed=3, perturb=True)
    net = net_from_cfg(emu, cfg)
    theta = pack_theta(net, params)
    L, Odim, A = cfg.history_len, cfg.obs_dim, cfg.num_actions
    need = emu.dtqn_forward_workspace_floats(ctypes.byref(net), 1)
    assert need > 0 and net.tiled == 0
    ws = np.zeros(need, dtype=np.float32)
    rng = np.random.default_rng(11)
    for n in (1, 20, 33, L):
        obs = rng.uniform(-1, 1, size=(1, n, Odim)).astype(np.float32)
        act = rng.integers(0, A, size=(1, n, 1))
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs), torch.as_tensor(act, dtype=torch.long)).numpy()[0, -1]
        ctx_h = np.zeros(L * Odim * 4 + L, dtype=np.uint8)
        ctx_h[:L * Odim * 4].view(np.float32).reshape(L, Odim)[:n] = obs[0]
        ctx_h[L * Odim * 4:][:n] = act[0, :, 0]
        for workspace in (ws, None):
            ctx_d = np.zeros_like(ctx_h)
            q_d = np.full((L, A), np.nan, dtype=np.float32)
            q_last = np.full(A, np.nan, dtype=np.float32)
            rc = emu.dtqn_actor_forward(ctypes.byref(net), ptr(theta), ptr(ctx_h), ptr(ctx_d), n, ptr(q_d), ptr(q_last),
                                        None if workspace is None else ptr(workspace), 0, 0, 0, None)
            assert rc == 0
            assert np.abs(q_last - ref).max() <= 1e-4, (n, workspace is None)
            assert np.array_equal(q_last, q_d[n - 1])
        assert not ws[emu.dtqn_td_xch_floats(ctypes.byref(net), 1):].any()      # hand-over flags lowered again
    assert emu.dtqn_actor_forward(ctypes.byref(net), ptr(theta), ptr(ctx_h), ptr(ctx_d), L + 1, ptr(q_d), ptr(q_last), None, 0, 0, 0, None) == B.DEFINES["DTQN_ERR_ARG"]


BAG = [
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=20, bag_size=5),
    dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=4, history_len=70, discrete=True, vocab_sizes=9, action_dim=8, bag_size=7,
         num_layers=1),
    dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=2, history_len=12, action_dim=4, bag_size=12, gate="gru", pos="sin"),
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, history_len=20, bag_size=6, identity=True),
]


@pytest.mark.parametrize("kw", BAG)
def test_forward_with_a_bag(emu, kw):
    """DTQN.forward with bag_obss / bag_actions (dtqn.py:201-214): the bag entries are embedded (own actions, no roll), the final
    hidden states cross-attend to them without a mask, and the Q head reads [hidden | attended]."""
    cfg = O.NetCfg(**kw)
    net = net_from_cfg(emu, cfg)
    assert net.tiled == 1 and net.bag_size == cfg.bag_size       # bag networks always take the row-block tiled path
    params = O.init_params(cfg, seed=4, perturb=True)
    assert params["ffn.0.weight"].shape == (cfg.inner_embed_size, 2 * cfg.inner_embed_size)
    theta = pack_theta(net, params)
    rng = np.random.default_rng(6)
    for n in sorted({1, cfg.history_len // 2 + 1, cfg.history_len}):
        Bn = 3
        draw = lambda m: (rng.integers(0, cfg.vocab_sizes, size=(Bn, m, cfg.obs_dim)) if cfg.discrete
                          else rng.uniform(-1, 1, size=(Bn, m, cfg.obs_dim)).astype(np.float32))
        obs, bag_obs = draw(n), draw(cfg.bag_size)
        act = rng.integers(0, cfg.num_actions, size=(Bn, n, 1))
        bag_act = rng.integers(0, cfg.num_actions, size=(Bn, cfg.bag_size, 1))
        ot = torch.long if cfg.discrete else torch.float32
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs, dtype=ot), torch.as_tensor(act, dtype=torch.long),
                            bag_obss=torch.as_tensor(bag_obs, dtype=ot), bag_actions=torch.as_tensor(bag_act, dtype=torch.long)).numpy()
        q = np.full((Bn, n, cfg.num_actions), np.nan, dtype=np.float32)
        ws = np.zeros(emu.dtqn_forward_workspace_floats(ctypes.byref(net), Bn), dtype=np.float32)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        u8 = lambda a: np.ascontiguousarray(a.reshape(Bn, -1), dtype=np.uint8)
        rc = emu.dtqn_forward_bag(ctypes.byref(net), ptr(theta), ptr(f32(obs)), ptr(u8(act)), ptr(f32(bag_obs)), ptr(u8(bag_act)), Bn, n,
                                  ptr(q), ptr(ws), 0, 0, 0, None)
        assert rc == 0
        assert np.abs(q - ref).max() <= 1e-4, (n, np.abs(q - ref).max())
    # the bag-less entry refuses a bag network
    assert emu.dtqn_forward_tiled(ctypes.byref(net), ptr(theta), ptr(f32(obs)), ptr(u8(act)), Bn, n, ptr(q), ptr(ws), None) != 0


def test_bag_configurations_outside_the_kernels_are_refused(emu):
    for kw in (dict(bag_size=80), dict(inner_embed_size=32, num_heads=4)):
        kw = {**dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=20, bag_size=5), **kw}
        with pytest.raises(NotImplementedError):
            net_from_cfg(emu, O.NetCfg(**kw))
