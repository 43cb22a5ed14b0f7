"""Width-padded networks (include/dtqn_hip.h, DtqnNet.d_real): a d_model the kernels are not instantiated for (48, 80, 96, 160 ...) runs at
the next width that is, with whole extra heads; padded entries are zero and stay zero.  On the test-only emulation: the TD update against
the oracle at the REAL width, the reference-shaped state_dict of the module, an agent that trains.  (`-m gpu` twins: tests/test_gpu_td.py,
tests/test_gpu_surface.py.)"""
import ctypes

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from helpers import make_td_case, check_td_updates, net_from_cfg, pack_theta, padding_mask

# (reference constructor arguments, run, (padded d_model, padded heads))
PADDED = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=2, history_len=20), dict(batch=2, T=30, mask=-5, tuf=2), (64, 8)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=6, num_layers=1, history_len=30, discrete=True, vocab_sizes=9, pos="sin"),
     dict(batch=2, T=40, mask=8), (128, 8)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=80, num_heads=5, num_layers=1, history_len=12, gate="gru"), dict(batch=2, T=20, mask=-5), (128, 8)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=40, num_heads=5, num_layers=1, history_len=70, identity=True), dict(batch=2, T=90, mask=-5), (64, 8)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=160, num_heads=5, num_layers=1, history_len=16), dict(batch=2, T=24, mask=-5), (256, 8)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=12, num_layers=1, history_len=10, gate="gru", identity=True, pos="none"),
     dict(batch=2, T=16, mask=-5), (64, 16)),
    # head widths that are not instantiated: every head padded to the next one that is (12 -> 16, 24 -> 32, 6 -> 8, 20 -> 32), softmax scale
    # of the real width; the last one also gets an extra head
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=4, num_layers=2, history_len=20), dict(batch=2, T=30, mask=-5, tuf=2), (64, 4)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=4, num_layers=1, history_len=30, discrete=True, vocab_sizes=9, pos="sin", gate="gru"),
     dict(batch=2, T=40, mask=8), (128, 4)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=8, num_layers=1, history_len=70, identity=True), dict(batch=2, T=90, mask=-5), (64, 8)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=60, num_heads=3, num_layers=1, history_len=12), dict(batch=2, T=20, mask=-5), (128, 4)),
    # one head of 96 -> 128 columns (round 6: head width 128 exists for contexts up to 64 rows)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=96, num_heads=1, num_layers=1, history_len=12), dict(batch=2, T=20, mask=-5), (128, 1)),
    # widths 16 / 32 beyond the row counts their whole-sequence kernels exist for: padded to 64 columns with extra heads
    (dict(obs_dim=3, num_actions=3, inner_embed_size=32, num_heads=4, num_layers=1, history_len=40), dict(batch=2, T=50, mask=-5), (64, 8)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=32, num_heads=1, num_layers=1, history_len=8), dict(batch=2, T=14, mask=-5), (64, 2)),
    # next to an action embedding (round 5): a token is [action embedding | observation embedding] (dtqn.py:192), the real columns stay a prefix
    (dict(obs_dim=3, num_actions=4, inner_embed_size=48, num_heads=6, num_layers=2, history_len=20, action_dim=8), dict(batch=2, T=30, mask=-5, tuf=2), (64, 8)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=4, num_layers=1, history_len=30, discrete=True, vocab_sizes=9, pos="sin", gate="gru", action_dim=4),
     dict(batch=2, T=40, mask=8), (128, 4)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=40, num_heads=5, num_layers=1, history_len=70, identity=True, action_dim=12), dict(batch=2, T=90, mask=-5), (64, 8)),
]


# ... with dropout (round 5): the keep masks of the embedding and of the feed-forward output are keyed by (row, REAL column) -- TlDrop.dw in
# dtqn_tiled.hip --, so a padded network drops exactly the units the reference-shaped network of the oracle drops
PADDED_DROPOUT = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=2, history_len=20, dropout=0.1), dict(batch=2, T=30, mask=-5, tuf=2), (64, 8)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=4, num_layers=1, history_len=30, discrete=True, vocab_sizes=9, pos="sin", gate="gru", action_dim=4,
          dropout=0.2), dict(batch=2, T=40, mask=8), (128, 4)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=40, num_heads=5, num_layers=1, history_len=70, identity=True, dropout=0.15), dict(batch=2, T=90, mask=-5), (64, 8)),
]


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


def four_slice_family(kw, padded):
    """dtqn_limits.h, dtqn_ws_lite: a padded width of 64 at head width 8 / 16 / 32, residual gate, post-LN, no dropout, context <= 64 rows runs on
    the four-slice whole-sequence kernels (round 5); everything else on the row-block kernels."""
    return (padded[0] == 64 and padded[0] // padded[1] in (8, 16, 32) and kw.get("gate", "res") == "res" and not kw.get("identity", False)
            and kw.get("dropout", 0.0) == 0.0 and kw["history_len"] <= 64)


@pytest.mark.parametrize("family", ["default", "row-block"])
@pytest.mark.parametrize("kw,run,padded", PADDED)
def test_td_update_of_a_width_padded_network(emu, kw, run, padded, family, monkeypatch):
    lite = four_slice_family(kw, padded)
    if family == "row-block":
        if not lite:
            pytest.skip("the default family of this shape is the row-block one already")
        monkeypatch.setenv("DTQN_WS_LITE_OFF", "1")       # A/B knob read by dtqn_net_init: the row-block kernels of rounds 1-4
        lite = False
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=19, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"], tuf=run.get("tuf", 10_000))
    assert (net.d_real, net.heads_real) == (cfg.inner_embed_size, cfg.num_heads) and (net.d_model, net.num_heads) == padded
    hd = cfg.inner_embed_size // cfg.num_heads
    assert net.tiled == (0 if lite else 1) and net.hd_real == hd and net.head_dim == next(w for w in (4, 8, 16, 32, 64, 128) if w >= hd)
    if lite:
        assert net.lp == 64 and eng.net.tiled == 0 and eng.row_split == 4
    assert padding_mask(net).sum() > 0
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3)      # incl.: padded gradient entries == 0, padded parameters stay 0


@pytest.mark.parametrize("kw,run,padded", [c for c in PADDED if four_slice_family(c[0], c[2])])
def test_pipelined_td_update_of_a_width_padded_network(emu, kw, run, padded):
    """... and as the timed flavour: policy passes as four slices, the next update's target pass inside the backward launch."""
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=29, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"], tuf=run.get("tuf", 10_000))
    assert net.tiled == 0 and net.d_real == cfg.inner_embed_size
    assert eng.enable_pipeline(lambda: 0) and eng._pipe["ride"]
    w = check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=4, pipelined=True)
    assert w["pipeline"]["used"] >= 1 and w["pipeline"]["used"] + w["pipeline"]["inline"] == 4
    assert int(eng.xflags.sum()) == 0 and int(eng._next_xflags.sum()) == 0


@pytest.mark.parametrize("kw,run,padded", PADDED_DROPOUT)
def test_td_update_of_a_width_padded_network_with_dropout(emu, kw, run, padded):
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=23, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"], tuf=run.get("tuf", 10_000))
    assert (net.d_real, net.d_model, net.num_heads) == (cfg.inner_embed_size,) + padded and net.dropout > 0
    eng.td.dropout_seed = 9876
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


@pytest.mark.parametrize("family", ["default", "row-block"])
def test_forward_on_context_prefixes(emu, family, monkeypatch):
    from helpers import ptr
    if family == "row-block":
        monkeypatch.setenv("DTQN_WS_LITE_OFF", "1")
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=2, history_len=20, pos="sin")
    net = net_from_cfg(emu, cfg)
    assert net.tiled == (1 if family == "row-block" else 0)
    params = O.init_params(cfg, seed=3, perturb=True)
    theta = torch.from_numpy(pack_theta(net, params))
    rng = np.random.default_rng(1)
    ws = torch.empty(emu.dtqn_forward_workspace_floats(ctypes.byref(net), 3))
    for n in (1, 2, 10, 20):
        obs = torch.tensor(rng.uniform(-1, 1, (3, n, 3)).astype(np.float32))
        act = torch.tensor(rng.integers(0, 3, (3, n)).astype(np.uint8))
        q = torch.full((3, n, 3), float("nan"))
        if net.tiled:
            assert emu.dtqn_forward_tiled(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 3, n, ptr(q), ptr(ws), None) == 0
        else:       # one workgroup per sequence on the whole 64-row tile (the only inference flavour of these shapes besides four slices)
            assert emu.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 3, n, ptr(q), None) == 0
        with torch.no_grad():
            ref = O.forward(params, cfg, obs, act.long().unsqueeze(-1)).numpy()
        assert np.abs(q.numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), n


def test_what_padding_does_not_cover_is_refused(emu):
    ok = dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=1, history_len=10)
    assert B.make_net(emu, **ok).d_real == 48
    for bad in (dict(inner_embed_size=280, num_heads=2),   # head width 140: beyond the widest attention instantiation (128)
                dict(inner_embed_size=140, num_heads=2, history_len=100),   # head width 70 -> 128: one head's tile fits LDS up to 64 rows only
                dict(inner_embed_size=240, num_heads=6),   # six heads of 40 -> 64 columns each: 384 > 256
                dict(bag_size=4),
                dict(inner_embed_size=272, num_heads=17)):
        with pytest.raises(NotImplementedError):
            B.make_net(emu, **{**ok, **bad})
    again = B.make_net(emu, **ok)                     # a padded struct initialised again keeps the caller's width
    assert emu.dtqn_net_init(ctypes.byref(again)) == 0 and (again.d_real, again.d_model, again.heads_real, again.num_heads) == (48, 64, 6, 8)
    twin = B.DtqnNet()                                 # ... and so does its row-block twin
    assert emu.dtqn_net_tiled_twin(ctypes.byref(again), ctypes.byref(twin)) == 0 and (twin.d_real, twin.d_model, twin.hd_real) == (48, 64, 8)


@pytest.mark.parametrize("heads", [6, 4])                # 6 heads of 8: two extra heads; 4 heads of 12: every head padded to 16
def test_module_speaks_the_reference_shapes(emu, heads):
    """DTQN(inner_embed_size=48, num_heads=6): state_dict() has the reference's shapes, a reference-shaped state_dict loads, the forward
    equals the oracle at width 48, and everything outside the real entries of the flat buffer is zero."""
    from dtqn_amd.networks.dtqn import DTQN
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=heads, num_layers=2, history_len=12, gate="gru")
    m = DTQN(3, 3, 8, 0, 48, heads, 2, 12, gate="gru", _test_lib=emu)
    m._allow_cpu = True
    params = O.init_params(cfg, seed=7, perturb=True)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in params.items()}
    pad = padding_mask(m.net)
    assert pad.any() and not m.flat[:m.net.n_trainable][torch.from_numpy(pad)].any()       # fresh module: zero padding (LayerNorm gammas included)
    real = ~pad
    assert m.flat[:m.net.n_trainable][torch.from_numpy(real)].abs().sum() > 0
    m.load_state_dict(params)
    assert not m.flat[:m.net.n_trainable][torch.from_numpy(pad)].any()
    back = m.state_dict()
    for k, v in params.items():
        assert torch.equal(back[k], v), k
    rng = np.random.default_rng(2)
    obs = torch.tensor(rng.uniform(-1, 1, (2, 9, 3)).astype(np.float32))
    act = torch.zeros(2, 9, 1, dtype=torch.long)
    with torch.no_grad():
        ref = O.forward(params, cfg, obs, act)
    assert (m(obs, act) - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))
    # a second module takes the first one's state_dict (DqnAgent.target_update, dqn.py:208-210)
    m2 = DTQN(3, 3, 8, 0, 48, heads, 2, 12, gate="gru", _test_lib=emu)
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2.flat, m.flat)


def test_agent_trains_at_a_padded_width(emu):
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.random import set_global_seed
    from test_emu_agent import make_agent
    env = envs.make("DiscreteCarFlag-v0")
    set_global_seed(4, env)
    agent = make_agent(emu, env, batch=2, L=8, D=48, H=6, tuf=2)
    net = agent.policy_network.net
    assert (net.d_real, net.d_model) == (48, 64)
    runpy.prepopulate(agent, 600, [env])
    theta0 = agent.policy_network.flat.clone()
    agent.context_reset(env.reset())
    for _ in range(3):
        if runpy.step(agent, env, Constant(0.3)):
            agent.replay_buffer.flush(); agent.context_reset(env.reset())
        agent.train()
    assert agent.num_train_steps == 3 and np.isfinite(agent.td_errors.mean())
    pad = torch.from_numpy(padding_mask(net))
    for flat in (agent.policy_network.flat, agent.target_network.flat, agent.engine.adam_m, agent.engine.adam_v):
        assert not flat[:net.n_trainable][pad].any()
    assert not torch.equal(theta0, agent.policy_network.flat)
    assert tuple(agent.policy_network.state_dict()["transformer_layers.0.attention.in_proj_weight"].shape) == (144, 48)


def test_pad_and_unpad_are_inverse_on_every_tensor_kind():
    """_binding.pad_param / unpad_param over random (heads, head width, padded heads, padded head width): unpad(pad(x)) == x bit for bit,
    the padding is zero, and along the head-structured axes every head's real entries sit in front of its own block."""
    import types
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=100, deadline=None, derandomize=True)
    @given(st.integers(1, 6), st.integers(1, 24), st.integers(0, 3), st.sampled_from([4, 8, 16, 32, 64]), st.booleans())
    def run(H, hd, extra_heads, hdp, as_torch):
        if hdp < hd:
            hdp = next(w for w in (4, 8, 16, 32, 64, 128) if w >= hd)
        Hp, D, Dp = H + extra_heads, H * hd, (H + extra_heads) * hdp
        if D == Dp:
            return
        net = types.SimpleNamespace(d_real=D, heads_real=H, hd_real=hd, num_heads=Hp, head_dim=hdp, d_model=Dp)
        rng = np.random.default_rng(H * 100 + hd)
        cases = {"transformer_layers.0.attention.in_proj_weight": ((3 * D, D), (3 * Dp, Dp)),
                 "transformer_layers.0.attention.in_proj_bias": ((3 * D,), (3 * Dp,)),
                 "transformer_layers.0.attention.out_proj.weight": ((D, D), (Dp, Dp)),
                 "transformer_layers.0.ffn.0.weight": ((4 * D, D), (4 * Dp, Dp)),
                 "position_embedding.position_encoding": ((1, 5, D), (1, 5, Dp)),
                 "ffn.2.weight": ((3, D), (3, Dp))}
        for key, (rs, ps) in cases.items():
            x = rng.standard_normal(rs).astype(np.float32) + 3.0             # no zeros in the real part
            x = torch.from_numpy(x) if as_torch else x
            p = B.pad_param(net, key, x, ps)
            assert tuple(p.shape) == ps and int((p != 0).sum()) == int(np.prod(rs))
            back = B.unpad_param(net, key, p, rs)
            assert (torch.equal(back, x) if as_torch else np.array_equal(back, x)), key
        w = np.arange(3 * D * D, dtype=np.float32).reshape(3 * D, D) + 1
        p = B.pad_param(net, "transformer_layers.0.attention.in_proj_weight", w, (3 * Dp, Dp))
        blk, h, i = 2, H - 1, hd - 1                                        # v block, last real head, last real column of the head
        assert np.array_equal(p[blk * Dp + h * hdp + i, :D], w[blk * D + h * hd + i])
    run()


def test_beyond_latency_mode_a_four_slice_only_shape_trains_on_its_row_block_twin(emu, monkeypatch):
    """dtqn_td_prefers_tiled for the shapes of dtqn_ws_lite: where dtqn_td_row_split does not say 4 (large batches; here DTQN_ROW_SPLIT=0), the
    update runs on the row-block twin of the net -- same theta layout -- while the caller's net stays whole-sequence."""
    monkeypatch.setenv("DTQN_ROW_SPLIT", "0")
    for kw in (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=1, history_len=20),
               dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=2, num_layers=1, history_len=20)):
        cfg = O.NetCfg(**kw)
        net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=31, batch=2, T=30, n_eps=6, mask=-5)
        assert net.tiled == 0 and eng.actor_net.tiled == 0 and eng.net.tiled == 1 and eng.row_split == 1
        check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
