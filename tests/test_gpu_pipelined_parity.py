"""-m gpu: the kernels bench.py TIMES, under the oracle directly.

bench.py's headline runs DtqnAgent.train() with the device sampler: dtqn_td_update_pipelined =
dtqn_forward_kernel<64, 1, 8, 8, false, 4, true, false> (policy passes, four 16-row slices, window draw in-kernel) +
dtqn_backward_kernel<64, 1, 8, 8, false, 4, false, false, true> (data-gradient chain + the NEXT update's target pass) + the weight-
gradient / optimizer launches.  Every other oracle comparison of the suite feeds host-drawn windows (set_indices), which takes the
un-pipelined two-slice launch.  Here the engine draws its own windows, the oracle batch is rebuilt from the (episode, start) pairs the
kernels left behind, and Q x 3, gradients, statistics and the Adam step are compared exactly as in tests/test_gpu_td.py
(reference: dtqn/agents/dtqn.py:215-265).  The measured margins go to gpurun_out/parity_report.json."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dtqn_oracle as O

from conftest import GOLDEN
from helpers import check_td_updates, flat_from_params, make_td_case, net_from_cfg, oracle_batch, pack_theta, parity_report

pytestmark = pytest.mark.gpu

CFG1 = dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


@pytest.mark.parametrize("tuf,scale", [(10_000, 0.1), (2, 0.1), (10_000, 1.0)])
def test_pipelined_update_cfg1_full_size_vs_oracle(lib, tuf, scale):
    """BASELINE config 1 at full size (batch 32), the update as bench.py times it.  scale 0.1: init_weights-scale matrices, |Q| < 1,
    so the Q bound is the absolute 1e-4 of north_star; scale 1: the std-0.2 stress weights of the other TD cases.  tuf = 2: every
    second update syncs the target network, which invalidates the pass computed ahead (it then runs inline)."""
    cfg = O.NetCfg(**CFG1)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=32, T=200, n_eps=40, mask=-5, tuf=tuf, device="cuda",
                                               test_lib=False, weight_scale=scale)
    assert eng.row_split == 4 and eng.enable_pipeline(lambda: 0) and eng._pipe["ride"]
    w = check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=6, pipelined=True,
                         report_as=f"pipelined_cfg1_tuf{tuf}_scale{scale}")
    if scale == 0.1:
        assert w["q_abs_max"] < 1.0 and w["q_abs_err"] <= 1e-4, w          # absolute
    assert w["pipeline"]["used"] >= (5 if tuf > 6 else 2), w["pipeline"]      # the pass computed ahead WAS what the loss read
    assert int(eng.xflags.sum()) == 0 and int(eng._next_xflags.sum()) == 0


def test_pipelined_update_head_width_16_vs_oracle(lib):
    """The second instantiation pair of the pipelined update (<64, 1, 16, 8, false, 4, ...>: --heads 4 at d_model 64)."""
    cfg = O.NetCfg(**{**CFG1, "num_heads": 4, "num_actions": 4})
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=22, batch=24, T=120, n_eps=40, mask=-5, history=30, device="cuda", test_lib=False)
    assert eng.row_split == 4 and eng.enable_pipeline(lambda: 0) and eng._pipe["ride"]
    w = check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=4, pipelined=True, report_as="pipelined_d64_hd16")
    assert w["pipeline"]["used"] >= 3


@pytest.mark.parametrize("width,heads,shape", [(64, 2, (64, 2, 32, 0)), (48, 6, (64, 8, 8, 48)), (48, 4, (64, 4, 16, 48)), (40, 2, (64, 2, 32, 40))])
def test_pipelined_update_head_width_32_and_padded_widths_vs_oracle(lib, width, heads, shape):
    """Round 5: `--heads 2` and the width-padded shapes of d_model 64 on the four-slice kernels instead of the
    row-block ones (dtqn_limits.h, dtqn_ws_lite): <64, 1, 32, 8, false, 4, ...> (a head spans two column tiles of the delta
    epilogue) and the PAD instantiations (LayerNorm statistics over the real columns, softmax scale of the real head width), as the
    update bench.py times, at batch 32.  check_td_updates also demands that no padded entry takes a gradient or moves."""
    cfg = O.NetCfg(**{**CFG1, "inner_embed_size": width, "num_heads": heads})
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=24, batch=32, T=200, n_eps=40, mask=-5, tuf=3, device="cuda", test_lib=False)
    assert (net.d_model, net.num_heads, net.head_dim, net.d_real) == shape and net.tiled == 0 and eng.net.tiled == 0
    assert eng.row_split == 4 and eng.enable_pipeline(lambda: 0) and eng._pipe["ride"]
    w = check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=5, pipelined=True, report_as=f"pipelined_w{width}_h{heads}")
    assert w["pipeline"]["used"] >= 2 and w["pipeline"]["used"] + w["pipeline"]["inline"] == 5
    assert int(eng.xflags.sum()) == 0 and int(eng._next_xflags.sum()) == 0


def test_a_large_batch_of_a_four_slice_only_shape_trains_on_its_row_block_twin(lib):
    """dtqn_td_prefers_tiled: beyond latency mode (batch 64) a width-padded network's update runs on the row-block twin; acting stays on
    the whole-sequence net.  Same oracle checks."""
    cfg = O.NetCfg(**{**CFG1, "inner_embed_size": 48, "num_heads": 6, "num_layers": 1})
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=25, batch=64, T=120, n_eps=80, mask=-5, device="cuda", test_lib=False)
    assert net.tiled == 0 and eng.actor_net.tiled == 0 and eng.net.tiled == 1 and eng.net.d_real == 48
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


def test_four_slice_forward_through_the_staged_update_vs_oracle(lib, monkeypatch):
    """DTQN_FWD_SLICES=4 routes dtqn_td_forward (host-drawn windows) onto the four-slice kernels: the same instantiation checked
    with the reference-stream indices of the other TD cases."""
    monkeypatch.setenv("DTQN_FWD_SLICES", "4")
    cfg = O.NetCfg(**CFG1)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=32, T=200, n_eps=40, mask=-5, device="cuda", test_lib=False)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3, report_as="fwd_slices4_cfg1")
    assert int(eng.xflags.sum()) == 0


def test_pipelined_update_against_the_references_own_numbers_G1(lib):
    """G1 (the reference's own train() on config 1) through the pipelined kernels.  The replay holds golden window b as episode b
    (length L: the start draw is always 0), so whatever multiset of windows the in-kernel draw picks, row b of the three Q tensors
    must be the reference's Q of window ep[b] to the absolute 1e-4; the gradient of the drawn batch is checked against the oracle
    conditional on the engine's own ReLU / argmax pattern like everywhere else."""
    from dtqn_amd.learner import DeviceReplay, TdEngine
    from helpers import engine_probe
    z = np.load(os.path.join(GOLDEN, "G1_cfg1_td.npz"))
    cfg = O.NetCfg(**json.loads(str(z["cfg"])))
    seed, Bn, L = int(z["seed"]), int(z["B"]), cfg.history_len
    pol = O.init_params(cfg, seed=seed, perturb=True)
    tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
    net = net_from_cfg(lib, cfg)
    eng = TdEngine(net, Bn, lr=float(z["lr"]), gamma=float(z["gamma"]), history=int(z["history"]), tuf=int(z["tuf"]))
    eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, pol)))
    eng.theta_tgt.copy_(torch.from_numpy(pack_theta(net, tgt)))
    rep = DeviceReplay(Bn, L, cfg.obs_dim, float(z["mask"]), eng.device)
    obs = np.concatenate([z["batch0_obss"], z["batch0_next_obss"][:, -1:]], axis=1).astype(np.float32)
    act = np.concatenate([z["batch0_actions"][:, :, 0], z["batch0_next_actions"][:, -1:, 0]], axis=1).astype(np.uint8)
    rep.obs.copy_(torch.from_numpy(obs)); rep.actions.copy_(torch.from_numpy(act))
    rep.rewards.copy_(torch.from_numpy(z["batch0_rewards"][:, :, 0].astype(np.float32)))
    rep.dones.copy_(torch.from_numpy(z["batch0_dones"][:, :, 0].astype(np.uint8)))
    rep.ep_len.fill_(L)
    assert eng.enable_pipeline(lambda: 0) and eng._pipe["ride"]
    worst = 0.0
    for it in range(3):                       # update 0: target pass inline; 1, 2: the pass the previous backward launch carried
        eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, pol)))          # same parameters every time: Q rows stay comparable
        eng.adam_m.zero_(); eng.adam_v.zero_()
        eng.sample_in_forward(Bn, -1, 4321)
        eng.update(rep)
        torch.cuda.synchronize()
        idx = eng._idx_dev.cpu().numpy()
        eps, starts = idx[0], idx[1]
        assert (starts == 0).all() and ((eps >= 0) & (eps < Bn)).all() and len(set(eps.tolist())) > Bn // 3
        q3 = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :cfg.num_actions]
        for w, name in enumerate(("q_all", "q_next_pol", "q_next_tgt")):
            err = float(np.abs(q3[w] - z[name][eps]).max())
            worst = max(worst, err)
            assert err <= 1e-4, (it, name, err)
        # gradient of the drawn multiset vs the oracle (conditional on the engine's pattern)
        ot = torch.float32
        batch = O.Batch(obss=torch.as_tensor(z["batch0_obss"][eps], dtype=ot), actions=torch.as_tensor(z["batch0_actions"][eps], dtype=torch.long),
                        rewards=torch.as_tensor(z["batch0_rewards"][eps], dtype=torch.float32),
                        next_obss=torch.as_tensor(z["batch0_next_obss"][eps], dtype=ot),
                        next_actions=torch.as_tensor(z["batch0_next_actions"][eps], dtype=torch.long),
                        dones=torch.as_tensor(z["batch0_dones"][eps], dtype=torch.long))
        probe = engine_probe(cfg, net, eng)
        grads, _ = O.td_gradients(pol, tgt, cfg, batch, float(z["gamma"]), int(z["history"]), probe)
        keys = O.trainable_keys(cfg)
        ref_flat = flat_from_params(net, grads, keys)
        got = eng.grad.cpu().numpy()
        cerr = float(np.abs(got - ref_flat).max() / np.abs(ref_flat).max())
        # (windows are drawn with replacement: a kink that sits within rounding in one golden window counts once per copy in the batch)
        assert cerr <= 2e-4 and probe.get("relu_flips", 0) <= 8 and probe.get("argmax_flips", 0) <= 2, (it, cerr, probe.get("relu_flips"), probe.get("argmax_flips"))
        assert probe.get("max_flip_preact", 0.0) <= 2e-5 * max(1.0, float(np.abs(z["q_all"]).max())), probe.get("max_flip_preact")
    assert eng._pipe["used"] == 2 and eng._pipe["inline"] == 1
    parity_report("pipelined_G1", {"q_abs_err_vs_reference": worst, "pipeline": {"used": 2, "inline": 1}})


def test_side_stream_flavour_cfg5_shapes_vs_oracle(lib):
    """BASELINE config 5 shapes (ctx 256, d_model 256: row-block kernels): the pipelined update's second-stream flavour -- the next
    update's target pass on a side stream beside this update's backward kernels -- against the oracle on the drawn windows."""
    cfg = O.NetCfg(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, history_len=256, discrete=True, vocab_sizes=22)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=2, T=260, n_eps=5, mask=21, device="cuda", test_lib=False)
    assert eng.net.tiled == 1 and eng.enable_pipeline(lambda: 0) and not eng._pipe["ride"]
    w = check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3, pipelined=True, report_as="pipelined_cfg5_side_stream")
    assert w["pipeline"]["used"] >= 2


def test_gru_nets_do_not_take_the_pipelined_update(lib):
    """The pass-ahead design covers residual gates (dtqn_td_fwd_slices4_ok); a GRU-gated config-1 net keeps the staged launches the
    other TD cases check (tests/test_gpu_td.py CASES[6])."""
    cfg = O.NetCfg(**{**CFG1, "gate": "gru"})
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=16, T=200, n_eps=30, mask=-5, device="cuda", test_lib=False)
    assert not eng.enable_pipeline(lambda: 0)
