"""Kernel-logic tests on the CPU for the full TD update (forward x3, loss, backward, weight
gradients, reduce, clip + Adam) on the test-only HIP emulation, against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from helpers import make_td_case, check_td_updates


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


CASES = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8), dict(batch=4, T=12, mask=-5)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, history_len=20, action_dim=4), dict(batch=3, T=30, mask=-5, history=7, tuf=2)),
    (dict(obs_dim=10, num_actions=5, inner_embed_size=32, num_heads=2, history_len=12, discrete=True, vocab_sizes=9), dict(batch=5, T=20, mask=8)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, identity=True, pos="sin"), dict(batch=4, T=12, mask=-5, tuf=3)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=32, num_heads=4, history_len=30, discrete=True, vocab_sizes=22, action_dim=8, pos="none", identity=True),
     dict(batch=2, T=40, mask=21)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, gate="gru"), dict(batch=4, T=12, mask=-5, tuf=2)),
    (dict(obs_dim=10, num_actions=5, inner_embed_size=32, num_heads=4, history_len=20, discrete=True, vocab_sizes=9, gate="gru", identity=True, action_dim=4, pos="sin"),
     dict(batch=3, T=30, mask=8)),
]


@pytest.mark.parametrize("kw,run", CASES)
def test_td_update_small_variants(emu, kw, run):
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=9, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3)


@pytest.mark.parametrize("kw,run", [CASES[1], CASES[2], CASES[6]])
def test_td_update_split_weight_gradients(emu, kw, run, monkeypatch):
    """Small batches take the one-launch weight-gradient kernel (16 x 32 tiles straight into grad); this forces the
    large-batch path (64 x 64 tiles per batch split + dtqn_td_reduce) on the same cases."""
    import ctypes
    monkeypatch.setenv("DTQN_WGRAD_DIRECT", "0")
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=9, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    assert emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), run["batch"]) == 0
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    monkeypatch.delenv("DTQN_WGRAD_DIRECT")
    assert emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), run["batch"]) == 1
    assert emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), 2048 // net.lp + 1) == 0


@pytest.mark.parametrize("kw,run", [CASES[0], CASES[2], CASES[5]])
def test_td_update_one_call(emu, kw, run):
    """dtqn_td_update, the single call DtqnAgent.train() issues (forward, backward, weight gradients, clip + Adam)."""
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=9, mask=run["mask"],
                                               history=run.get("history"), tuf=2)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3, one_call=True)


def test_td_update_cfg1_size(emu):
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=5, batch=4, T=200, n_eps=8, mask=-5)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


TILED = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=20), dict(batch=2, T=30, mask=-5)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=70, action_dim=4, pos="sin"),
     dict(batch=2, T=90, mask=-5, history=9, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=2, num_layers=1, history_len=12, discrete=True, vocab_sizes=9, action_dim=8),
     dict(batch=3, T=20, mask=8)),
    # identity-reordered layers (transformer.py:86-101): the LayerNorms sit on the branches
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, num_layers=2, history_len=70, identity=True),
     dict(batch=2, T=90, mask=-5, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=8, num_layers=1, history_len=12, discrete=True, vocab_sizes=9, action_dim=4,
          identity=True, pos="sin"), dict(batch=3, T=20, mask=8, history=5)),
    # GRU gates (gates.py:26-31), shared by both layers
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, num_layers=2, history_len=20, gate="gru"),
     dict(batch=2, T=30, mask=-5, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=8, num_layers=2, history_len=70, discrete=True, vocab_sizes=9, action_dim=4,
          gate="gru", identity=True), dict(batch=2, T=90, mask=8, history=30)),
    # d_model 128: the GEMM kernels read the fragment-major weight copies (dtqn_td_wpack, round 6); target sync every second update
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, num_layers=2, history_len=20), dict(batch=2, T=30, mask=-5, tuf=2)),
]


@pytest.mark.parametrize("kw,run", TILED)
def test_td_update_tiled_path(emu, kw, run, monkeypatch):
    """The row-block tiled training path (BASELINE configs 4 / 5) forced on small shapes: same checks as the
    whole-sequence kernels (Q x3, conditional gradients, statistics, Adam step, target sync)."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=33, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    assert net.tiled == 1 and net.lp % 64 == 0
    assert (emu.dtqn_td_wpack_floats(ctypes.byref(net)) > 0) == (cfg.inner_embed_size % 128 == 0) == hasattr(eng, "wpack_pol")
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3 if cfg.inner_embed_size % 128 == 0 else 2)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("kw,run", [TILED[0], TILED[7]])
def test_td_update_tiled_fused_layer_kernels(emu, kw, run, fused, monkeypatch, capfd):
    """Round 6: a post-LN residual layer behind its attention is ONE launch forward (tl_layer_kernel: out-projection, LayerNorm 1,
    feed-forward, LayerNorm 2, and the Q head on the last layer) and one backward (tl_chain_bwd_kernel: LayerNorm-2 backward, feed-forward
    backward, LayerNorm-1 backward, gate mask, dO = da W_o).  Both against the oracle, next to the separate launches they replace
    (DTQN_LAYER_FUSE=0 / DTQN_BWD_CHAIN=0), and the launch trace says which kernels ran."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_FFN_ROWS", "64")            # the chain kernel exists for 64-row workgroups (the small-batch rule picks 32)
    monkeypatch.setenv("DTQN_LAYER_FUSE", fused)
    monkeypatch.setenv("DTQN_BWD_CHAIN", fused)
    monkeypatch.setenv("DTQN_TL_TRACE", "1")
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=33, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    err = capfd.readouterr().err
    names = {"layer": "tl_layer_kernel" in err, "chain": "tl_chain_bwd_kernel" in err, "ffn": "tl_ffn_kernel" in err,
             "ffn_bwd": "tl_ffn_bwd_kernel" in err, "ln_bwd": "tl_layernorm_bwd_kernel" in err, "qhead": "tl_qhead_kernel" in err}
    if fused == "1":
        assert names == {"layer": True, "chain": True, "ffn": False, "ffn_bwd": False, "ln_bwd": False, "qhead": False}, names
    else:
        assert names == {"layer": False, "chain": False, "ffn": True, "ffn_bwd": True, "ln_bwd": True, "qhead": True}, names


@pytest.mark.parametrize("table", ["1", "0"])
@pytest.mark.parametrize("adim", [0, 4])
def test_td_update_tiled_embedding_product_table(emu, table, adim, monkeypatch, capfd):
    """Round 6: discrete observations of a covered row-block network (d_model 128) are embedded in the TD forward from the product table
    P[j][v] = T[v] W_e[:, slot j]^T that dtqn_td_wpack rewrites with the weight copies (tl_embed_table_kernel: O gathered rows per token);
    DTQN_EMBED_TABLE=0 keeps the matrix product (tl_embed_kernel).  Both against the oracle, with and without an action embedding beside it,
    over a target sync (the target's table follows theta_tgt)."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_EMBED_TABLE", table)
    monkeypatch.setenv("DTQN_EMBED_QKV", "1")            # (by default only launches of two rounds and more take the fused form)
    monkeypatch.setenv("DTQN_TL_TRACE", "1")
    cfg = O.NetCfg(obs_dim=6, num_actions=5, inner_embed_size=128, num_heads=8, num_layers=1, history_len=20, discrete=True, vocab_sizes=9,
                   action_dim=adim)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=11, batch=3, T=30, n_eps=6, mask=8, tuf=2)
    assert net.tiled == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3)
    err = capfd.readouterr().err
    assert ("tl_embed_table_kernel" in err) == (table == "1") and ("tl_launch tl_embed_kernel" in err) == (table == "0")
    # ... and with the table, layer 0's q | k | v projection rides in the embedding launch (d_model 128 / 256): no tl_wide_kernel launch at all
    # in this one-layer network; DTQN_EMBED_QKV=0 keeps them apart with bit-identical Q
    assert ("tl_wide_kernel" in err) == (table == "0")
    if table == "1":
        q_fused = eng.q3.clone()
        monkeypatch.setenv("DTQN_EMBED_QKV", "0")
        net2, oracle2, host2, eng2, rep2 = make_td_case(emu, cfg, seed=11, batch=3, T=30, n_eps=6, mask=8, tuf=2)
        check_td_updates(cfg, net2, oracle2, host2, eng2, rep2, n_updates=3)
        err2 = capfd.readouterr().err
        assert "tl_wide_kernel" in err2 and "tl_embed_table_kernel<0, false>" in err2
        assert torch.equal(eng2.q3, q_fused)


@pytest.mark.parametrize("ctx,batch", [(48, 4), (96, 2)])
def test_td_update_tiled_packed_rows_of_the_unsaved_passes(emu, ctx, batch, monkeypatch, capfd):
    """Round 6: the q | k | v projection and the fused layer tail walk the LIVE rows of policy(o') and target(o') 64 at a time (TlPack: a
    workgroup's rows run on from one sequence into the next; L = 48 / 96 in records of 64 / 128 rows), the training third stays on
    (sequence, row block).  Against the oracle, and bit-identical in Q to the unpacked walk (DTQN_PACK_ROWS=0)."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_FFN_ROWS", "64")
    cfg = O.NetCfg(obs_dim=6, num_actions=5, inner_embed_size=128, num_heads=8, num_layers=2, history_len=ctx, discrete=True, vocab_sizes=9)
    qs = {}
    for packed in ("0", "1"):
        monkeypatch.setenv("DTQN_PACK_ROWS", packed)
        monkeypatch.setenv("DTQN_TL_TRACE", "1")
        net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=13, batch=batch, T=ctx + 10, n_eps=6, mask=8, tuf=2)
        assert net.tiled == 1 and net.lp > ctx
        check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
        qs[packed] = eng.q3.clone()
        capfd.readouterr()
    assert torch.equal(qs["1"], qs["0"])
    # the packed launches are shorter: 3 B LPB / 64 workgroups unpacked, B LPB / 64 + 2 B L / 64 packed (seen through the grid the shim is given)
    assert emu.dtqn_debug_last_packed_blocks() == batch * (net.lp // 64) + 2 * batch * ctx // 64


@pytest.mark.parametrize("ctx", [300, 512])
def test_td_update_contexts_beyond_256(emu, ctx):
    """north_star's bound is a 512-step context.  The row-block kernels take records of up to 512 rows as long as one head's q | k | v | dO
    tile fits LDS: head width 16 here (68 floats per row: 139 KB at 512 rows).  One sequence, against the oracle."""
    cfg = O.NetCfg(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=ctx)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=9, batch=1, T=ctx + 8, n_eps=3, mask=-5)
    assert net.tiled == 1 and net.lp == (ctx + 63) // 64 * 64
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=1)


def test_td_update_contexts_beyond_512_or_too_wide_heads_are_refused(emu):
    from dtqn_amd import _binding as Bd
    with pytest.raises(Exception):
        Bd.make_net(emu, obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=513)
    with pytest.raises(Exception):                                      # head width 32 at 512 rows: 270 KB tile
        Bd.make_net(emu, obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=2, num_layers=1, history_len=400)


def test_td_update_tiled_next_layers_projection_rides_in_the_layer_launch(emu, monkeypatch, capfd):
    """Round 6: at d_model 128 / 256 the fused layer launch of layer l also runs layer l + 1's q | k | v projection on the tile its LayerNorm 2
    leaves (tl_layer_kernel<..., TAIL = 2>): a two-layer forward launches tl_wide_kernel once (layer 0) instead of twice.  Against the oracle,
    and bit-identical in Q to the separate launch (DTQN_QKV_FUSE=0)."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_FFN_ROWS", "64")
    monkeypatch.setenv("DTQN_TL_TRACE", "1")
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, num_layers=2, history_len=20)
    qs, wide, tails = {}, {}, {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("DTQN_QKV_FUSE", fuse)
        net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=19, batch=2, T=30, n_eps=6, mask=-5, tuf=2)
        capfd.readouterr()
        check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
        err = capfd.readouterr().err
        qs[fuse], wide[fuse], tails[fuse] = eng.q3.clone(), err.count("tl_launch (tl_wide_kernel"), err.count("true, 2>)") + err.count("false, 2>)")
    assert torch.equal(qs["0"], qs["1"])
    assert wide["0"] == 2 * wide["1"] > 0 and tails["0"] == 0 and tails["1"] == wide["1"]


def test_td_update_tiled_lds_weight_gradients(emu, monkeypatch):
    """Row-block network of d_model 128 on the LARGE-batch weight-gradient path (forced at a small batch): the layer matrices and the
    first head matrix through dtqn_wgrad_lds_kernel (128 x 128 tiles, operands staged through LDS), the embedding and the last head
    matrix through dtqn_wgrad_kernel, both into the same splits; against the oracle like every other case."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_WGRAD_DIRECT", "0")
    cfg = O.NetCfg(obs_dim=6, num_actions=5, inner_embed_size=128, num_heads=8, num_layers=2, history_len=40, discrete=True, vocab_sizes=9)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=5, batch=3, T=50, n_eps=6, mask=8, tuf=2)
    assert net.tiled == 1 and emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), 3) == 0
    assert eng.n_split == emu.dtqn_td_wgrad_splits(ctypes.byref(net), 3) == 3
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    monkeypatch.setenv("DTQN_WGRAD_LDS", "0")           # the same update with every job on the 64 x 64 kernel
    net2, oracle2, host2, eng2, rep2 = make_td_case(emu, cfg, seed=5, batch=3, T=50, n_eps=6, mask=8, tuf=2)
    check_td_updates(cfg, net2, oracle2, host2, eng2, rep2, n_updates=1)


SPLIT = [
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50),
    dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50, gate="gru", action_dim=4),
    dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=55, action_dim=8, pos="sin"),
    dict(obs_dim=6, num_actions=5, inner_embed_size=128, num_heads=8, num_layers=1, history_len=50, discrete=True, vocab_sizes=9),
]


@pytest.mark.parametrize("slices", ["1", "4", "4-unfused"])
@pytest.mark.parametrize("kw", SPLIT)
def test_td_update_row_split(emu, kw, slices, monkeypatch):
    """Latency mode: two workgroups per sequence (rows 0-31 / 32-63) with the K|V and dK|dV hand-over between them.
    Same checks as one workgroup per sequence, including a short history window that lives entirely in the upper slice.
    Four slices with residual gates: the backward launch also carries the weight-gradient workgroups (dtqn_td_wgrad_is_fused),
    "4-unfused" keeps them in their own launch."""
    import ctypes
    monkeypatch.setenv("DTQN_ROW_SPLIT", slices[0])         # "1": two slices in both kernels; "4": four in the backward
    monkeypatch.setenv("DTQN_WGRAD_FUSED", "0" if slices == "4-unfused" else "1")      # opt-in (measured slower on the GPU)
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=8, batch=3, T=80, n_eps=7, mask=-5 if not kw.get("discrete") else 8,
                                               history=None if kw["num_heads"] == 8 else 11, tuf=2)
    assert eng.row_split == (2 if slices == "1" else 4) and net.lp == 64
    fused = emu.dtqn_td_wgrad_is_fused(ctypes.byref(net), ctypes.byref(eng.td))
    assert fused == (1 if slices == "4" and kw.get("gate") != "gru" else 0)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    assert int(eng.xflags.sum()) == 0                      # every hand-over flag was lowered again, every event counter reset


@pytest.mark.parametrize("split", ["0", "1"])
def test_in_kernel_window_draw_equals_sample_kernel(emu, split, monkeypatch):
    """dtqn_td_forward with sample_in_kernel draws exactly the windows dtqn_replay_sample draws for the same
    (seed, step counter), leaves them in ep_idx / start for the backward, and produces the same update."""
    monkeypatch.setenv("DTQN_ROW_SPLIT", split)
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=50)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=4, batch=4, T=90, n_eps=9, mask=-5)
    eng.step_counter[1] = 7
    eng.sample_on_device(rep, 9, 3, 12345)
    ref = (eng.ep_idx.clone(), eng.start.clone())
    assert 3 not in ref[0].tolist() and len(set(ref[0].tolist())) > 1
    eng.forward_backward(rep)
    grad_ref = eng.grad.clone()
    eng.ep_idx.zero_(); eng.start.zero_()
    eng.sample_in_forward(9, 3, 12345)
    eng.forward_backward(rep)
    assert torch.equal(eng.ep_idx, ref[0]) and torch.equal(eng.start, ref[1])
    assert torch.equal(eng.grad, grad_ref)


@pytest.mark.parametrize("split,heads", [("4", 8), ("0", 4), ("tiled", 4)])
def test_pad_rows_with_overflowing_scores_stay_out_of_the_gradient(emu, split, heads, monkeypatch):
    """Regression: the padded query rows of a window (context 50 in a 64-row tile) carry bias-only activations; with
    large in-projection biases their recomputed attention probabilities overflow (their saved log-sum-exp is 0).  They
    must contribute exactly nothing: a 1 M-step training run once died of inf * 0 = NaN in dQ of a pad row."""
    if split == "tiled":
        monkeypatch.setenv("DTQN_FORCE_TILED", "1")       # the row-block tiled kernels share the attention backward
    else:
        monkeypatch.setenv("DTQN_ROW_SPLIT", split)
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=heads, num_layers=1, history_len=50)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=5, batch=2, T=90, n_eps=5, mask=-5)
    assert net.tiled == (1 if split == "tiled" else 0)
    # q / k biases of +-12 per column: pad rows (x = position row only -> after LN a fixed vector) get |q.k| * scale * log2(e) >> 128
    tab = B.param_table(net)
    off, shape = tab["transformer_layers.0.attention.in_proj_bias"]
    D = cfg.inner_embed_size
    bias = np.zeros(3 * D, dtype=np.float32)
    bias[:D] = 12.0
    bias[D:2 * D] = 12.0
    for th, params in ((eng.theta_pol, oracle.pol), (eng.theta_tgt, oracle.tgt)):
        th[off:off + 3 * D] = torch.from_numpy(bias)
        params["transformer_layers.0.attention.in_proj_bias"] = torch.from_numpy(bias.copy())
    eps, starts = host.sample_indices(2)
    eng.set_indices(eps, starts)
    eng.forward_backward(rep)
    assert bool(torch.isfinite(eng.grad).all())
    # scores of +-1e3 in front of the softmax: the one parity case whose Q tolerance is relative to |Q|max (measured 2e-4 absolute)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=1, q_rel=True)


DROPOUT_CASES = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, dropout=0.1), dict(batch=4, T=12, mask=-5)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, history_len=20, action_dim=4, dropout=0.25), dict(batch=3, T=30, mask=-5, history=7)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, dropout=0.1), dict(batch=2, T=60, mask=-5, n_eps=10)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, gate="gru", dropout=0.2), dict(batch=4, T=12, mask=-5)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=32, num_heads=4, history_len=20, identity=True, pos="sin", dropout=0.15), dict(batch=3, T=30, mask=-5)),
]


@pytest.mark.parametrize("kw,run", DROPOUT_CASES)
def test_td_update_with_dropout(emu, kw, run, monkeypatch):
    """dropout > 0 (dtqn.py:105,196; transformer.py:34,41): embedding, attention-probability and FFN-output dropout in the
    two train-mode forwards, none in the target forward; the backward recomputes the keep masks.  The oracle evaluates the
    same counter-based hash, so Q x3, gradients, statistics and the Adam step are compared as without dropout."""
    if kw["inner_embed_size"] == 64:
        monkeypatch.setenv("DTQN_ROW_SPLIT", "4")
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=run.get("n_eps", 9),
                                               mask=run["mask"], history=run.get("history"))
    eng.td.dropout_seed = 12345
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


TILED_DROPOUT = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=20, dropout=0.1), dict(batch=3, T=30, mask=-5, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=4, num_layers=1, history_len=70, discrete=True, vocab_sizes=9, action_dim=8,
          dropout=0.25), dict(batch=2, T=90, mask=8, history=30)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=2, num_layers=2, history_len=12, gate="gru", dropout=0.2), dict(batch=3, T=20, mask=-5)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, num_layers=1, history_len=20, identity=True, pos="sin", dropout=0.15),
     dict(batch=3, T=30, mask=-5)),
]


@pytest.mark.parametrize("ffn_bwd", ["1", "0"])
@pytest.mark.parametrize("kw,run", TILED_DROPOUT)
def test_td_update_with_dropout_on_the_row_block_path(emu, kw, run, ffn_bwd, monkeypatch):
    """The same on the row-block tiled kernels: the embedding epilogue, the attention kernels (one head per workgroup: global
    head index in the mask key), the fused feed-forward epilogue; in the backward the keep masks ride in the staging of the
    fused feed-forward backward (DTQN_FFN_BWD=1) or in a row kernel in front of the separate products (=0), and dL/dx0 takes the
    embedding mask before the table / position gradients."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_FFN_BWD", ffn_bwd)
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=23, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    assert net.tiled == 1 and net.dropout > 0
    eng.td.dropout_seed = 4321
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


def test_dropout_keep_rate_and_eval_mode(emu):
    """The keep masks drop a fraction p of the elements and scale the rest by 1 / (1 - p) (nn.Dropout's definition); an
    eval-mode forward (dtqn_forward, the target pass) is unaffected by the dropout setting."""
    import ctypes
    from helpers import net_from_cfg, pack_theta, ptr
    sp = O.DropSpec(0.3, 99, 4, 0)
    idx = np.arange(200_000).astype(np.uint64)
    for site in (O.DROP_EMB, O.DROP_ATTN, O.DROP_FFN):
        keep = O.drop_keep(sp, 3, site, 1, idx)
        assert abs(keep.mean() - 0.7) < 4 * np.sqrt(0.21 / idx.size)
    assert abs(sp.scale - 1 / 0.7) < 1e-6
    # independent across passes / sequences / steps
    a = O.drop_keep(sp, 3, 0, 0, idx[:4096]); b = O.drop_keep(O.DropSpec(0.3, 99, 5, 0), 3, 0, 0, idx[:4096])
    assert 0.35 < (a == b).mean() < 0.8
    kw = dict(obs_dim=3, num_actions=3, inner_embed_size=32, num_heads=4, history_len=20)
    cfg0, cfg1 = O.NetCfg(**kw), O.NetCfg(dropout=0.5, **kw)
    params = O.init_params(cfg0, seed=3, perturb=True)
    rng = np.random.default_rng(1)
    obs = rng.uniform(-1, 1, size=(2, 20, 3)).astype(np.float32)
    act = np.zeros((2, 20), dtype=np.uint8)
    outs = []
    for cfg in (cfg0, cfg1):
        net = net_from_cfg(emu, cfg)
        theta = pack_theta(net, params)
        q = np.full((2, 20, 3), np.nan, dtype=np.float32)
        assert emu.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 2, 20, ptr(q), None) == 0
        outs.append(q)
    assert np.array_equal(outs[0], outs[1])


BAG_TD = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=20, bag_size=5), dict(batch=3, T=30, mask=-5, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=4, num_layers=1, history_len=70, discrete=True, vocab_sizes=9, action_dim=8,
          bag_size=7), dict(batch=2, T=90, mask=8)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=2, num_layers=1, history_len=12, action_dim=4, bag_size=12, gate="gru"),
     dict(batch=3, T=20, mask=-5, history=6)),
    # dropout: the bag attention drops attention weights too (its own site of the mask key); bag embeddings are not dropped
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, num_layers=1, history_len=20, bag_size=6, dropout=0.2, action_dim=4),
     dict(batch=3, T=30, mask=-5)),
    # identity-reordered layers: no closing LayerNorm, the working memory is the last layer's stream
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, num_layers=2, history_len=20, bag_size=6, identity=True, pos="sin"),
     dict(batch=3, T=30, mask=-5, tuf=2)),
]


@pytest.mark.parametrize("kw,run", BAG_TD)
def test_td_update_with_a_bag(emu, kw, run):
    """TD update of a bag network (dtqn.py:201-214 inside DtqnAgent.train, dtqn.py:191-284): the same bag serves the three forwards;
    gradients of the cross-attention, of the [D][2D] head and of the embeddings through BOTH the context and the bag tokens."""
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=35, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    assert net.tiled == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


@pytest.mark.parametrize("kw,run", BAG_TD[:2])
def test_td_update_with_a_bag_split_weight_gradients(emu, kw, run, monkeypatch):
    """The same through the large-batch weight-gradient path (tiles per batch split + dtqn_td_reduce)."""
    monkeypatch.setenv("DTQN_WGRAD_DIRECT", "0")
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=36, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    assert not emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), run["batch"])
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


@pytest.mark.parametrize("rows", ["32", "64"])
@pytest.mark.parametrize("kw,run", [TILED[0], TILED[4], BAG_TD[1]])
def test_td_update_tiled_rows_per_workgroup(emu, kw, run, rows, monkeypatch):
    """The GEMM kernels of the row-block path (linear, dY W, fused feed-forward) with 32- and with 64-row workgroups (the launchers
    pick by launch size: small test shapes would always take 32)."""
    monkeypatch.setenv("DTQN_FORCE_TILED", "1")
    monkeypatch.setenv("DTQN_FFN_ROWS", rows)
    monkeypatch.setenv("DTQN_GEMM_ROWS", rows)
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=37, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000))
    assert net.tiled == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


def test_training_on_the_row_block_twin(emu, monkeypatch):
    """dtqn_td_prefers_tiled / dtqn_net_tiled_twin: a D = 128 residual post-LN net with 64-row contexts trains on the row-block twin
    of the caller's net (same theta layout, tiled records) while inference keeps the caller's net.  The policy asks for it beyond
    latency mode (batch > 42); DTQN_TRAIN_TILED=1 forces it at a batch the emulation finishes quickly."""
    cfg = O.NetCfg(obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=8, num_layers=1, history_len=50, action_dim=8)
    net = B.make_net(emu, obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=8, num_layers=1, history_len=50, action_dim=8)
    assert net.tiled == 0 and emu.dtqn_td_prefers_tiled(ctypes.byref(net), 2) == 0 and emu.dtqn_td_prefers_tiled(ctypes.byref(net), 64) == 1
    for kw in (dict(inner_embed_size=64), dict(gate="gru"), dict(identity=True)):     # shapes the policy leaves alone
        other = B.make_net(emu, **{**dict(obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=8, num_layers=1, history_len=50), **kw})
        assert emu.dtqn_td_prefers_tiled(ctypes.byref(other), 64) == 0, kw
    # a short context at this width sits on the same 64-row instantiation (dtqn_limits.h) and gets the same answer
    short = B.make_net(emu, obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=8, num_layers=1, history_len=20)
    assert short.lp == 64 and emu.dtqn_td_prefers_tiled(ctypes.byref(short), 64) == 1
    monkeypatch.setenv("DTQN_TRAIN_TILED", "1")
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=41, batch=2, T=60, n_eps=5, mask=-5, tuf=2)
    assert net.tiled == 0 and eng.net.tiled == 1 and eng.actor_net.tiled == 0
    assert eng.net.n_theta == net.n_theta and B.param_table(eng.net) == B.param_table(net)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    obs = torch.zeros(1, 7, cfg.obs_dim)
    q = eng.forward(obs, torch.zeros(1, 7, 1, dtype=torch.uint8))            # whole-sequence inference entry on the caller's net
    assert q.shape == (1, 7, cfg.num_actions) and torch.isfinite(q).all()
    monkeypatch.setenv("DTQN_TRAIN_TILED", "0")
    _, _, _, eng0, _ = make_td_case(emu, cfg, seed=41, batch=2, T=60, n_eps=5, mask=-5)
    assert eng0.net.tiled == 0


@pytest.mark.parametrize("ffn_bwd", ["0", "1", "chain", "unfused"])
def test_td_update_tiled_path_width_256(emu, ffn_bwd, monkeypatch, capfd):
    """D = 256 takes the two-chunk / two-column-block step sequences of the fused row-block kernels (tl_wide, tl_layer / tl_ffn, and with
    DTQN_FFN_BWD=1 the fused feed-forward backward that is off by default at this width; "chain": tl_chain_bwd_kernel, DTQN_BWD_CHAIN256=1;
    "unfused": every launch of rounds 1-5 -- DTQN_LAYER_FUSE=0, DTQN_HEAD_FUSE=0); everything else in this file runs D = 64 / 128."""
    if ffn_bwd == "chain":
        monkeypatch.setenv("DTQN_BWD_CHAIN256", "1")
    elif ffn_bwd == "unfused":
        monkeypatch.setenv("DTQN_LAYER_FUSE", "0")
        monkeypatch.setenv("DTQN_HEAD_FUSE", "0")
    else:
        monkeypatch.setenv("DTQN_FFN_BWD", ffn_bwd)
    monkeypatch.setenv("DTQN_TL_TRACE", "1")
    cfg = O.NetCfg(obs_dim=3, num_actions=4, inner_embed_size=256, num_heads=8, num_layers=1, history_len=12, action_dim=8)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=43, batch=2, T=20, n_eps=5, mask=-5)
    assert net.tiled == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=1)
    err = capfd.readouterr().err
    assert ("tl_chain_bwd_kernel" in err) == (ffn_bwd == "chain")
    assert ("tl_layer_kernel" in err) == (ffn_bwd != "unfused") and ("tl_head_bwd_kernel" in err) == (ffn_bwd == "unfused")


def test_bag_network_counts_both_token_sets_for_the_one_launch_weight_gradients(emu):
    """The embedding job of a bag network contracts the context tokens AND the bag entries.  With B * LP <= 2048 < 2 * B * LP the
    one-launch weight-gradient kernel (whose tiles hold at most 2048 / 16 token units) must not be chosen -- found on the GPU at
    B = 32 when the tiles went to 16 waves, where the second token set was silently dropped."""
    import ctypes
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=50, bag_size=5)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=35, batch=17, T=80, n_eps=24, mask=-5)
    assert net.tiled == 1 and 17 * net.lp <= 2048 < 2 * 17 * net.lp
    assert emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), 17) == 0 and emu.dtqn_td_wgrad_is_direct(ctypes.byref(net), 16) == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    # B = 16: 2 * 16 * 64 tokens = exactly what a one-launch tile holds (eight units per wave on all 16 waves)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=36, batch=16, T=80, n_eps=24, mask=-5)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


@pytest.mark.parametrize("heads", [8, 4])
def test_forward_in_parts_and_four_row_slices(emu, heads, monkeypatch):
    """dtqn_td_forward_part (round 4): the three passes launched in parts leave exactly what the one launch leaves, and the four
    16-row slices per sequence of the pipelined update (K | V of the lower rows handed from every slice to every slice above it)
    pass the same oracle parity as the two-slice forward -- cfg-1 network, in-kernel window draw with an explicit step."""
    import ctypes
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=heads, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=6, batch=3, T=120, n_eps=8, mask=-5)
    assert eng.row_split == 4 and emu.dtqn_td_fwd_slices4_ok(eng._net_ref) == 1
    n, r, t = eng._net_ref, rep.view_ref, eng._td_ref
    eng.sample_in_forward(8, 3, 99)
    eng.step_counter[1] = 4
    assert emu.dtqn_td_forward(n, r, t, None) == 0
    q_whole, idx_whole = eng.q3.clone(), eng._idx_dev.clone()      # (the activation records also hold pad rows nobody reads: not compared)
    for slices in (2, 4):
        for parts in (((0, 3),), ((0, 2), (2, 1)), ((2, 1), (1, 1), (0, 1))):
            eng.q3.zero_(); eng._idx_dev.zero_()
            for p0, np_ in parts:
                assert emu.dtqn_td_forward_part(n, r, t, p0, np_, slices, 4, None) == 0      # explicit draw step = the counter's value
            assert torch.equal(eng._idx_dev, idx_whole)
            if slices == 2:
                assert torch.equal(eng.q3, q_whole), (slices, parts)
            else:       # four slices: same numbers up to the summation order of the 16-row items
                assert (eng.q3 - q_whole).abs().max() <= 2e-5 * max(1.0, float(q_whole.abs().max()))
    # a different explicit step draws different windows
    assert emu.dtqn_td_forward_part(n, r, t, 0, 3, 4, 5, None) == 0
    assert not torch.equal(eng._idx_dev, idx_whole)
    # pass ranges are validated
    assert emu.dtqn_td_forward_part(n, r, t, 2, 2, 4, 4, None) != 0 and emu.dtqn_td_forward_part(n, r, t, 0, 1, 3, 4, None) != 0
    # the whole update on the four-slice forward against the oracle (DTQN_FWD_SLICES=4 routes dtqn_td_forward there)
    monkeypatch.setenv("DTQN_FWD_SLICES", "4")
    eng.step_counter[1] = 0
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


# Shapes that dtqn_net_init used to accept and the first launch then refused (no whole-sequence instantiation), or that it refused
# outright (head_dim 4 / 64): they now run on the smallest instantiated row-tile count, or on the row-block tiled path.
# (kw, run, expected (tiled, lp))
ROUTED = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=8), dict(batch=3, T=14, mask=-5), (0, 16)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=20), dict(batch=2, T=30, mask=-5, tuf=2), (0, 32)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=8, action_dim=4), dict(batch=3, T=14, mask=-5), (0, 64)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=8, gate="gru"), dict(batch=2, T=14, mask=-5), (0, 16)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=128, num_heads=8, num_layers=1, history_len=10, discrete=True, vocab_sizes=9), dict(batch=2, T=16, mask=8), (0, 64)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=2, num_layers=1, history_len=20), dict(batch=2, T=30, mask=-5), (0, 64)),       # head_dim 32: the four-slice kernels (round 5; dtqn_limits.h dtqn_ws_lite)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=1, num_layers=2, history_len=20), dict(batch=2, T=30, mask=-5, tuf=2), (1, 64)),  # head_dim 64: agent_utils.py's default num_heads=1
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=16, num_layers=1, history_len=12, pos="sin"), dict(batch=2, T=20, mask=-5), (1, 64)),  # head_dim 4
    (dict(obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=2, num_layers=1, history_len=70, gate="gru", action_dim=8), dict(batch=2, T=90, mask=-5), (1, 128)),  # head_dim 64
    # head_dim 128 (round 6): `--heads 1 --in-embed 128`, agent_utils.py's default num_heads=1 at the next width; one head's tile fits LDS up to 64 rows
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=1, num_layers=1, history_len=20), dict(batch=2, T=30, mask=-5, tuf=2), (1, 64)),
]


@pytest.mark.parametrize("kw,run,where", ROUTED)
def test_td_update_routed_shapes(emu, kw, run, where):
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=17, batch=run["batch"], T=run["T"], n_eps=6, mask=run["mask"], tuf=run.get("tuf", 10_000))
    assert (net.tiled, net.lp) == where
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


def test_every_accepted_shape_has_kernels(emu):
    """dtqn_net_init is the one place that says what is covered: every (d_model, heads, context) it accepts runs a forward that
    matches the oracle; what it refuses, it refuses there (DTQN_ERR_CONFIG), not at the first launch."""
    import itertools
    from helpers import net_from_cfg, pack_theta, ptr
    accepted = 0
    for D, H, L in itertools.product((16, 32, 48, 64, 96, 128, 160, 256), (1, 2, 3, 4, 5, 6, 8, 16), (8, 40, 100)):
        if D % H:
            continue
        cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=D, num_heads=H, history_len=L, num_layers=1)
        try:
            net = net_from_cfg(emu, cfg)
        except Exception as e:
            assert "rc=1" in str(e)
            continue
        accepted += 1
        params = O.init_params(cfg, seed=3, perturb=True)
        theta = torch.from_numpy(pack_theta(net, params))
        rng = np.random.default_rng(D + H + L)
        obs = torch.tensor(rng.uniform(-1, 1, (2, L, 3)).astype(np.float32))
        act = torch.tensor(rng.integers(0, 3, (2, L)).astype(np.uint8))
        q = torch.full((2, L, 3), float("nan"))
        if net.tiled:
            ws = torch.empty(emu.dtqn_forward_workspace_floats(ctypes.byref(net), 2))
            rc = emu.dtqn_forward_tiled(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 2, L, ptr(q), ptr(ws), None)
        else:
            rc = emu.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 2, L, ptr(q), None)
        assert rc == 0, (D, H, L, net.tiled, net.lp)
        hd = D // H                 # width-padded (DtqnNet.d_real): widths and head widths without an instantiation, 16 / 32 beyond their row counts
        assert (net.d_real > 0) == (D not in (16, 32, 64, 128, 256) or hd not in (4, 8, 16, 32, 64, 128) or (D < 64 and net.d_model == 64)), (D, H, L)
        with torch.no_grad():
            ref = O.forward(params, cfg, obs, act.long().unsqueeze(-1)).numpy()
        assert np.abs(q.numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (D, H, L)
    assert accepted >= 45


@pytest.mark.parametrize("heads,tuf", [(8, 10_000), (4, 2), (2, 3)])      # (2 heads: head width 32, two column tiles per head in the delta epilogue)
def test_pipelined_update_vs_oracle(emu, heads, tuf):
    """The update exactly as DtqnAgent.train() issues it with the device sampler (dtqn_td_update_pipelined: in-kernel window draw,
    policy passes as four 16-row slices, the NEXT update's target pass inside this update's backward launch and used from the second
    update on) against the oracle on the windows the kernels drew: Q x 3, conditional gradients, statistics, the Adam step, target
    syncs (tuf = 2: a sync invalidates the pass computed ahead, which then runs inline)."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=heads, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=12, batch=3, T=120, n_eps=8, mask=-5, tuf=tuf)
    assert eng.enable_pipeline(lambda: 0) and eng._pipe["ride"]
    w = check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=4, pipelined=True)
    assert w["pipeline"]["used"] >= (3 if tuf > 4 else 1) and w["pipeline"]["used"] + w["pipeline"]["inline"] == 4
    assert int(eng.xflags.sum()) == 0 and int(eng._next_xflags.sum()) == 0


def test_host_drawn_update_on_a_pipelined_engine_keeps_the_step_mirror(emu):
    """dtqn_td_update through set_indices on an engine whose pipeline is enabled steps the device's optimizer counter; the
    host mirror that keys the in-kernel draw and predicts the hard target sync must follow, and what was computed ahead is dropped."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=50)
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=13, batch=3, T=120, n_eps=8, mask=-5, tuf=3)
    assert eng.enable_pipeline(lambda: 0)
    eng.sample_in_forward(8, 3, 77)
    eng.update(rep)                                   # pipelined: step 0 -> 1, target pass for step 1 launched ahead
    assert eng._pipe["steps"] == 1 and eng._pipe["ahead"] is not None
    eps, starts = host.sample_indices(3)
    eng.set_indices(eps, starts)
    eng.update(rep)                                   # host-drawn windows: the plain one-call update, step 1 -> 2
    assert eng._pipe["steps"] == 2 == int(eng.step_counter[1]) and eng._pipe["ahead"] is None
    eng.sample_in_forward(8, 3, 77)
    eng.update(rep)                                   # step 2 -> 3: the hard sync (tuf = 3) is predicted from the right step
    assert eng._pipe["steps"] == 3 == int(eng.step_counter[1]) and eng._pipe["ahead"] is None
    assert float(eng.read_stats()["target_synced"]) == 1.0
    assert torch.equal(eng.theta_tgt[:net.n_trainable], eng.theta_pol[:net.n_trainable])
    with pytest.raises(TypeError):
        eng.enable_pipeline(None)


def test_backward_stays_sliced_past_latency_mode(emu):
    """dtqn_td_row_split / dtqn_td_latency_mode (round 5): 43 ... 64 sequences at d_model 64 run the backward chain in four slices and 65 ...
    128 in two while the forward runs one workgroup per sequence, and the pipelined form is for latency mode only.  One update at batch
    43 against the oracle (whole-sequence forward records under the four-slice backward)."""
    from helpers import net_from_cfg as net_from_cfg_
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=50)
    net = net_from_cfg_(emu, cfg)
    policy = {b: (emu.dtqn_td_row_split(ctypes.byref(net), b), emu.dtqn_td_latency_mode(ctypes.byref(net), b)) for b in (32, 42, 43, 64, 65, 128, 129, 256)}
    assert policy == {32: (4, 1), 42: (4, 1), 43: (4, 0), 64: (4, 0), 65: (2, 0), 128: (2, 0), 129: (1, 0), 256: (1, 0)}
    wide = net_from_cfg_(emu, O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, num_layers=1, history_len=50))
    assert emu.dtqn_td_row_split(ctypes.byref(wide), 32) == 4 and emu.dtqn_td_row_split(ctypes.byref(wide), 64) == 1
    net, oracle, host, eng, rep = make_td_case(emu, cfg, seed=14, batch=43, T=80, n_eps=50, mask=-5)
    assert eng.row_split == 4 and not eng.enable_pipeline(lambda: 0)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=1, one_call=True)
