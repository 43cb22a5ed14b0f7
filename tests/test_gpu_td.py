"""-m gpu: the full HIP TD update (through the C ABI) vs the oracle and the reference's golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from conftest import GOLDEN
from helpers import make_td_case, check_td_updates, net_from_cfg, pack_theta, flat_from_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


CASES = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8), dict(batch=4, T=12, mask=-5)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, history_len=20, action_dim=4), dict(batch=3, T=30, mask=-5, history=7, tuf=2)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50), dict(batch=32, T=200, mask=-5, n_eps=40)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, identity=True, pos="sin", action_dim=8), dict(batch=16, T=200, mask=-5, n_eps=30, tuf=2)),
    (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9), dict(batch=8, T=50, mask=8, n_eps=20)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=64, num_heads=4, history_len=64, discrete=True, vocab_sizes=22, pos="none"), dict(batch=6, T=70, mask=21, n_eps=12)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, gate="gru"), dict(batch=16, T=200, mask=-5, n_eps=30, tuf=2)),
    (dict(obs_dim=10, num_actions=10, inner_embed_size=64, num_heads=8, history_len=50, discrete=True, vocab_sizes=9, gate="gru", identity=True, action_dim=8, pos="sin"),
     dict(batch=8, T=50, mask=8, n_eps=20)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, history_len=8, gate="gru"), dict(batch=4, T=12, mask=-5)),
    # row-block tiled training path (L > 64 or D > 128): BASELINE config 4 / 5 shapes and two in-between ones
    (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=128, discrete=True, vocab_sizes=12), dict(batch=4, T=140, mask=11, n_eps=8)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, history_len=256, discrete=True, vocab_sizes=22), dict(batch=2, T=260, mask=21, n_eps=5)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, history_len=100, action_dim=8, pos="sin", num_layers=3), dict(batch=5, T=150, mask=-5, n_eps=8, history=30, tuf=2)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=256, num_heads=16, history_len=50, num_layers=1), dict(batch=3, T=60, mask=-5, n_eps=6)),
    # variants the whole-sequence LDS tile set cannot hold run on the tiled path: GRU gates / identity layers at D >= 128
    (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9, gate="gru"),
     dict(batch=8, T=50, mask=8, n_eps=20, tuf=2)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, history_len=50, identity=True), dict(batch=8, T=200, mask=-5, n_eps=20)),
    (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=128, discrete=True, vocab_sizes=12, gate="gru", action_dim=8),
     dict(batch=3, T=140, mask=11, n_eps=8, history=40)),
    # (GRU + identity + discrete tokens at D = 128, L = 128 with the perturbed test weights is ill-conditioned: the fp32
    #  oracle itself sits 9e-4 from its fp64 evaluation there, as does the HIP path -- so that combination is tested on
    #  continuous observations at L = 50; its gradient sits at 2.3e-4 of the largest entry with the 16-grouped contraction order
    #  of the row-block GEMMs and within 2e-4 with the previous order: summation-order noise, so this one case gets 4e-4)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, history_len=50, gate="gru", identity=True, pos="sin"),
     dict(batch=4, T=200, mask=-5, n_eps=12, tuf=2, grad_rtol=4e-4)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, history_len=100, discrete=True, vocab_sizes=22, gate="gru", num_layers=1),
     dict(batch=2, T=120, mask=21, n_eps=5)),
]


# contexts of 257 .. 512 steps (north_star's bound; round 6): records of up to 512 rows where one head's q | k | v | dO tile fits LDS (head width <= 16)
CASES += [
    (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=384, discrete=True, vocab_sizes=12), dict(batch=2, T=400, mask=11, n_eps=4)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=8, history_len=512, num_layers=1), dict(batch=2, T=520, mask=-5, n_eps=4, tuf=2)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=64, num_heads=8, history_len=300, discrete=True, vocab_sizes=22, gate="gru", num_layers=1),
     dict(batch=2, T=310, mask=21, n_eps=4)),
]

# shapes dtqn_net_init places on a larger instantiated row-tile count or sends to the row-block path (dtqn_limits.h): short contexts,
# head_dim 32 / 64 / 4 (round 4; the same list runs on the emulation, tests/test_emu_td.py ROUTED)
CASES += [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=8), dict(batch=32, T=14, mask=-5, n_eps=40)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=2, history_len=8, action_dim=4), dict(batch=16, T=14, mask=-5, n_eps=30, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=128, num_heads=8, num_layers=1, history_len=10, discrete=True, vocab_sizes=9), dict(batch=8, T=16, mask=8, n_eps=20)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=2, num_layers=2, history_len=50), dict(batch=8, T=200, mask=-5, n_eps=20)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=1, num_layers=2, history_len=50), dict(batch=8, T=200, mask=-5, n_eps=20, tuf=2)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=16, num_layers=1, history_len=50, pos="sin"), dict(batch=4, T=200, mask=-5, n_eps=12)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=128, num_heads=2, num_layers=1, history_len=100, gate="gru", action_dim=8), dict(batch=3, T=120, mask=-5, n_eps=8)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=4, num_layers=1, history_len=128, discrete=True, vocab_sizes=22), dict(batch=2, T=140, mask=21, n_eps=5)),
    # width-padded networks (DtqnNet.d_real; the same mechanism on the emulation: tests/test_padded_width.py): 48 -> 64, 96 -> 128, 80 -> 128 (GRU),
    # 160 -> 256; check_td_updates also demands that no padded entry takes a gradient or moves
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=2, history_len=50), dict(batch=16, T=200, mask=-5, n_eps=30, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=6, num_layers=2, history_len=50, discrete=True, vocab_sizes=9, pos="sin"),
     dict(batch=8, T=60, mask=8, n_eps=20)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=80, num_heads=5, num_layers=1, history_len=100, gate="gru", identity=True), dict(batch=4, T=120, mask=-5, n_eps=10)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=160, num_heads=5, num_layers=1, history_len=50), dict(batch=4, T=200, mask=-5, n_eps=12)),
    # ... and padded head widths (12 -> 16, 24 -> 32, 20 -> 32 plus an extra head), softmax scale of the real width; width 32 beyond 32 rows
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=4, num_layers=2, history_len=50), dict(batch=16, T=200, mask=-5, n_eps=30, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=4, num_layers=1, history_len=100, discrete=True, vocab_sizes=9, pos="sin", gate="gru"),
     dict(batch=4, T=120, mask=8, n_eps=10)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=60, num_heads=3, num_layers=1, history_len=50, identity=True), dict(batch=4, T=200, mask=-5, n_eps=12)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=32, num_heads=4, num_layers=1, history_len=50), dict(batch=8, T=200, mask=-5, n_eps=20)),
    # ... and width padding next to an action embedding (round 5: the action columns come first in a token, dtqn.py:192)
    (dict(obs_dim=3, num_actions=4, inner_embed_size=48, num_heads=6, num_layers=2, history_len=50, action_dim=8), dict(batch=16, T=200, mask=-5, n_eps=30, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=4, num_layers=1, history_len=100, discrete=True, vocab_sizes=9, pos="sin", gate="gru", action_dim=4),
     dict(batch=4, T=120, mask=8, n_eps=10)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=40, num_heads=5, num_layers=1, history_len=50, identity=True, action_dim=12), dict(batch=4, T=200, mask=-5, n_eps=12)),
    # head width 128 (round 6): `--heads 1 --in-embed 128` (agent_utils.py's default num_heads=1), two heads at d_model 256, and a padded 96-wide head
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=1, num_layers=2, history_len=50), dict(batch=8, T=200, mask=-5, n_eps=20, tuf=2)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=256, num_heads=2, num_layers=1, history_len=50, discrete=True, vocab_sizes=9, gate="gru"), dict(batch=4, T=60, mask=8, n_eps=10)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=96, num_heads=1, num_layers=1, history_len=50), dict(batch=4, T=200, mask=-5, n_eps=12)),
]


@pytest.mark.parametrize("kw,run", CASES)
def test_td_update_vs_oracle(lib, kw, run):
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=run.get("n_eps", 9),
                                               mask=run["mask"], history=run.get("history"), tuf=run.get("tuf", 10_000),
                                               device="cuda", test_lib=False)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3, grad_rtol=run.get("grad_rtol", 2e-4))
    assert int(eng.xflags.sum()) == 0          # latency mode: every hand-over flag was lowered again


@pytest.mark.parametrize("kw,run", [CASES[2], CASES[6], CASES[9], CASES[13]])
def test_td_update_split_weight_gradients(lib, kw, run, monkeypatch):
    """The large-batch weight-gradient path (64 x 64 tiles per batch split + dtqn_td_reduce) forced on small batches,
    which otherwise take the one-launch kernel; and a batch large enough to take it by itself."""
    monkeypatch.setenv("DTQN_WGRAD_DIRECT", "0")
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=run.get("n_eps", 9),
                                               mask=run["mask"], history=run.get("history"), tuf=run.get("tuf", 10_000),
                                               device="cuda", test_lib=False)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)


@pytest.mark.parametrize("kw,run", [CASES[0], CASES[2], CASES[3], CASES[6], CASES[4]])
def test_td_update_one_call(lib, kw, run):
    """dtqn_td_update as the agent's train() calls it: one library call, four launches at small batches."""
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=run.get("n_eps", 9),
                                               mask=run["mask"], history=run.get("history"), tuf=2, device="cuda", test_lib=False)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=4, one_call=True)
    assert int(eng.xflags.sum()) == 0


def test_one_call_update_is_bit_reproducible(lib):
    """Two engines, same state, 20 updates each at BASELINE config 1 (latency mode, one-launch weight gradients):
    identical parameters, moments, statistics."""
    import random
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    outs = []
    for rep_i in range(2):
        net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=9, batch=32, T=200, n_eps=60, mask=-5, device="cuda", test_lib=False)
        random.seed(5)                      # host.sample_indices draws from Python's `random` like the reference
        for it in range(20):
            eps, starts = host.sample_indices(32)
            eng.set_indices(eps, starts)
            eng.update(rep)
        torch.cuda.synchronize()
        st = eng.read_stats()
        assert st["nonfinite"] == 0.0 and st["step"] == 20
        outs.append((eng.theta_pol.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.stats.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_td_update_batch_256(lib):
    """BASELINE config 2 (batch 256: one workgroup per sequence, split weight gradients) against the oracle."""
    import ctypes
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=3, batch=256, T=200, n_eps=300, mask=-5, device="cuda", test_lib=False)
    assert lib.dtqn_td_wgrad_is_direct(ctypes.byref(net), 256) == 0 and eng.row_split == 1
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=1)


def test_latency_mode_is_on_for_the_metric_config(lib):
    """BASELINE config 1 (batch 32, 64-row tile): four backward / two forward workgroups per sequence; batch 256: one."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    net = net_from_cfg(lib, cfg)
    import ctypes
    assert lib.dtqn_td_row_split(ctypes.byref(net), 32) == 4 and lib.dtqn_td_latency_mode(ctypes.byref(net), 32) == 1
    assert lib.dtqn_td_row_split(ctypes.byref(net), 256) == 1
    # past latency mode the backward alone stays sliced while its workgroups fit the chip at once (round 5)
    assert [(lib.dtqn_td_row_split(ctypes.byref(net), b), lib.dtqn_td_latency_mode(ctypes.byref(net), b)) for b in (64, 128)] == [(4, 0), (2, 0)]


@pytest.mark.parametrize("batch,slices", [(64, 4), (128, 2)])
def test_sliced_backward_under_whole_sequence_forward_vs_oracle(lib, batch, slices):
    """Batches past latency mode (dtqn_td_row_split 4 / 2 with dtqn_td_latency_mode 0): forward passes one workgroup per sequence,
    backward chain in four / two row slices; the update in one call, as DtqnAgent.train() issues it (no pipelined form there)."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=27, batch=batch, T=120, n_eps=batch + 20, mask=-5, device="cuda", test_lib=False)
    assert eng.row_split == slices and net.tiled == 0 and not eng.enable_pipeline(lambda: 0)
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2, one_call=True, report_as=f"sliced_backward_b{batch}")


def test_golden_G1_full_update(lib):
    """cfg 1 at full size against the numbers the reference itself produced (tests/golden/G1)."""
    from dtqn_amd.learner import DeviceReplay, TdEngine
    z = np.load(os.path.join(GOLDEN, "G1_cfg1_td.npz"))
    cfg = O.NetCfg(**json.loads(str(z["cfg"])))
    seed, Bn, L = int(z["seed"]), int(z["B"]), cfg.history_len
    pol = O.init_params(cfg, seed=seed, perturb=True)
    tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
    net = net_from_cfg(lib, cfg)
    eng = TdEngine(net, Bn, lr=float(z["lr"]), gamma=float(z["gamma"]), history=int(z["history"]), tuf=int(z["tuf"]))
    eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, pol)))
    eng.theta_tgt.copy_(torch.from_numpy(pack_theta(net, tgt)))
    # a replay whose episode b is exactly the golden window b (obs rows 0..L, using the +1 overlap)
    rep = DeviceReplay(Bn, L, cfg.obs_dim, float(z["mask"]), eng.device)
    obs = np.concatenate([z["batch0_obss"], z["batch0_next_obss"][:, -1:]], axis=1).astype(np.float32)
    act = np.concatenate([z["batch0_actions"][:, :, 0], z["batch0_next_actions"][:, -1:, 0]], axis=1).astype(np.uint8)
    rep.obs.copy_(torch.from_numpy(obs)); rep.actions.copy_(torch.from_numpy(act))
    rep.rewards.copy_(torch.from_numpy(z["batch0_rewards"][:, :, 0].astype(np.float32)))
    rep.dones.copy_(torch.from_numpy(z["batch0_dones"][:, :, 0].astype(np.uint8)))
    eng.set_indices(np.arange(Bn), np.zeros(Bn))
    eng.forward_backward(rep)
    q3 = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :cfg.num_actions]
    scale = 1.0          # absolute tolerance (north_star: 1e-4 fp32)
    for w, name in enumerate(("q_all", "q_next_pol", "q_next_tgt")):
        assert np.abs(q3[w] - z[name]).max() <= 1e-4 * scale, name
    keys = O.trainable_keys(cfg)
    # golden flat gradient (oracle key order) -> engine layout
    ref, off = {}, 0
    shapes = O.param_shapes(cfg)
    for k in keys:
        n = int(np.prod(shapes[k]))
        ref[k] = torch.from_numpy(z["grad0_flat"][off:off + n].reshape(shapes[k]).copy())
        off += n
    ref_flat = flat_from_params(net, ref, keys)
    got = eng.grad.cpu().numpy()
    gerr = np.abs(got - ref_flat).max()
    # 1.3 M ReLU decisions and 1600 argmax choices: two correct fp32 implementations can sit on different sides of a kink
    # whose pre-activation is ~1e-7 (helpers.check_td_updates).  So: (i) the oracle evaluated on the ENGINE's activation
    # pattern must agree with the engine to 2e-4 of max|g| and disagree with its own pattern in at most 2 ReLUs / 1 argmax;
    # (ii) when there is no disagreement at all, the engine must meet the reference's own gradient to the same bound.
    from helpers import engine_probe, oracle_batch  # noqa: F401
    from test_gpu_parity_holes import report
    probe = engine_probe(cfg, net, eng)
    ot = torch.float32
    batch = O.Batch(obss=torch.as_tensor(z["batch0_obss"], dtype=ot), actions=torch.as_tensor(z["batch0_actions"], dtype=torch.long),
                    rewards=torch.as_tensor(z["batch0_rewards"], dtype=torch.float32), next_obss=torch.as_tensor(z["batch0_next_obss"], dtype=ot),
                    next_actions=torch.as_tensor(z["batch0_next_actions"], dtype=torch.long), dones=torch.as_tensor(z["batch0_dones"], dtype=torch.long))
    cgrads, _ = O.td_gradients(pol, tgt, cfg, batch, float(z["gamma"]), int(z["history"]), probe)
    cond_flat = flat_from_params(net, cgrads, keys)
    flips = int(probe.get("relu_flips", 0)) + int(probe.get("argmax_flips", 0))
    cerr = np.abs(got - cond_flat).max() / np.abs(cond_flat).max()
    report("G1_gradient", {"unconditional_err_over_max": float(gerr / np.abs(ref_flat).max()), "conditional_err_over_max": float(cerr),
                           "relu_flips": int(probe.get("relu_flips", 0)), "argmax_flips": int(probe.get("argmax_flips", 0))})
    assert cerr <= 2e-4, cerr
    assert probe.get("relu_flips", 0) <= 2 and probe.get("argmax_flips", 0) <= 1, probe
    strict = gerr <= 2e-4 * np.abs(ref_flat).max()
    assert strict or flips > 0, (gerr / np.abs(ref_flat).max(), probe)
    assert gerr <= 0.05 * np.abs(ref_flat).max()
    eng.clip_adam()
    st = eng.read_stats()
    ref_stats = json.loads(str(z["stats"]))[0]
    for k, v in ref_stats.items():
        assert abs(st[k] - v) <= (2e-4 if strict or k != "grad_norm" else 2e-2) * max(1.0, abs(v)), (k, st[k], v)
    if strict:
        post = eng.theta_pol.cpu().numpy()[:net.n_trainable]
        ref_post, off = {}, 0
        for k in keys:
            n = int(np.prod(shapes[k]))
            ref_post[k] = torch.from_numpy(z["post0_flat"][off:off + n].reshape(shapes[k]).copy())
            off += n
        d = np.abs(post - flat_from_params(net, ref_post, keys))
        solid = np.abs(ref_flat) >= 1e-3 * np.abs(ref_flat).max()
        assert d[solid].max() <= 2e-6
        assert d.max() <= 2.002 * float(z["lr"])


@pytest.mark.parametrize("name", ["cfg4", "cfg5"])
def test_golden_G3_tiled_update(lib, name):
    """BASELINE configs 4 and 5 (L = 128 / 256, D = 128 / 256) on the row-block tiled training path against
    the reference's own numbers (tests/golden/G3): Q x3, the seven logged statistics + gradient norm of the
    first update, and the policy's Q-values after the Adam step."""
    from dtqn_amd.learner import DeviceReplay, TdEngine
    z = np.load(os.path.join(GOLDEN, "G3_cfg345_td.npz"))
    g = lambda k: z[f"{name}/{k}"]
    cfg = O.NetCfg(**json.loads(str(g("cfg"))))
    seed, Bn, L = int(g("seed")), int(g("B")), cfg.history_len
    pol = O.init_params(cfg, seed=seed, perturb=True)
    tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
    net = net_from_cfg(lib, cfg)
    assert net.tiled == 1
    eng = TdEngine(net, Bn, lr=float(g("lr")), gamma=float(g("gamma")), history=int(g("history")), tuf=int(g("tuf")))
    eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, pol)))
    eng.theta_tgt.copy_(torch.from_numpy(pack_theta(net, tgt)))
    rep = DeviceReplay(Bn, L, cfg.obs_dim, float(g("mask")), eng.device)
    obs = np.concatenate([g("batch0_obss"), g("batch0_next_obss")[:, -1:]], axis=1).astype(np.float32)
    act = np.concatenate([g("batch0_actions")[:, :, 0], g("batch0_next_actions")[:, -1:, 0]], axis=1).astype(np.uint8)
    rep.obs.copy_(torch.from_numpy(obs)); rep.actions.copy_(torch.from_numpy(act))
    rep.rewards.copy_(torch.from_numpy(g("batch0_rewards")[:, :, 0].astype(np.float32)))
    rep.dones.copy_(torch.from_numpy(g("batch0_dones")[:, :, 0].astype(np.uint8)))
    eng.set_indices(np.arange(Bn), np.zeros(Bn))
    eng.forward_backward(rep)
    q3 = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :cfg.num_actions]
    scale = max(1.0, np.abs(g("q_all")).max())      # std-0.2 stress weights: see tests/helpers.py
    for w, nm in enumerate(("q_all", "q_next_pol", "q_next_tgt")):
        assert np.abs(q3[w] - g(nm)).max() <= 1e-4 * scale, nm
    eng.clip_adam()
    st = eng.read_stats()
    ref_stats = json.loads(str(g("stats")))[0]
    for k, v in ref_stats.items():
        # a ReLU kink flip vs the reference moves the gradient norm by more than rounding (see helpers.check_td_updates)
        assert abs(st[k] - v) <= (2e-4 if k != "grad_norm" else 5e-3) * max(1.0, abs(v)), (k, st[k], v)
    # Q-values of the updated policy on the same batch
    eng.forward_backward(rep)
    q_after = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[0, :, :L, :cfg.num_actions]
    moved = np.abs(g("q_all_final") - g("q_all")).max()
    err = np.abs(q_after - g("q_all_final")).max()
    assert moved > 10 * err and err <= 2e-2 * scale, (moved, err)


@pytest.mark.parametrize("split,D,gate", [("4", 64, "res"), ("4", 128, "res"), ("1", 128, "res"), ("4", 64, "gru")])
def test_gradient_is_bit_reproducible_in_latency_mode(lib, split, D, gate, monkeypatch):
    """No atomics, fixed reduction and hand-over orders: the gradient of a fixed batch is bit-identical over many
    repetitions, whatever the inter-workgroup timing (a cheap detector for races in the row-slice hand-overs)."""
    monkeypatch.setenv("DTQN_ROW_SPLIT", split)
    cfg = O.NetCfg(obs_dim=3, num_actions=5, inner_embed_size=D, num_heads=8, history_len=50, gate=gate)
    Bn = 32 if D == 64 else 16
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=3, batch=Bn, T=120, n_eps=40, mask=-5, device="cuda", test_lib=False)
    assert eng.row_split == (4 if split == "4" else 2)
    eps, starts = host.sample_indices(Bn)
    eng.set_indices(eps, starts)
    eng.forward_backward(rep)
    torch.cuda.synchronize()
    ref, refq = eng.grad.clone(), eng.q3.clone()
    assert bool(torch.isfinite(ref).all())
    for it in range(60):
        eng.forward_backward(rep)
        if it % 10 == 9:
            torch.cuda.synchronize()
            assert torch.equal(eng.grad, ref) and torch.equal(eng.q3, refq), it
    assert int(eng.xflags.sum()) == 0


@pytest.mark.parametrize("D,discrete", [(64, False), (128, True)])
def test_fused_weight_gradients_in_the_backward_launch(lib, D, discrete, monkeypatch):
    """Opt-in DTQN_WGRAD_FUSED=1 (latency mode): the backward launch carries the weight-gradient workgroups (write-through records,
    event counters, sc1 loads).  Same checks against the oracle as the separate launch, the gradient equals the separate launch's to
    rounding (8- vs 16-wave summation order), it is bit-reproducible over repetitions under uneven timing, and the counters end at zero."""
    import ctypes
    kw = dict(obs_dim=6 if discrete else 3, num_actions=5, inner_embed_size=D, num_heads=8, history_len=50, num_layers=2 if D == 64 else 1)
    if discrete:
        kw.update(discrete=True, vocab_sizes=9)
    cfg = O.NetCfg(**kw)
    Bn = 32 if D == 64 else 16
    mask = 8 if discrete else -5
    grads = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("DTQN_WGRAD_FUSED", fused)
        net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=3, batch=Bn, T=120, n_eps=40, mask=mask, device="cuda", test_lib=False)
        assert eng.row_split == 4
        assert lib.dtqn_td_wgrad_is_fused(ctypes.byref(net), ctypes.byref(eng.td)) == int(fused) and eng.wgrad_fused == (fused == "1")
        eps, starts = host.sample_indices(Bn)
        eng.set_indices(eps, starts)
        eng.forward_backward(rep)
        torch.cuda.synchronize()
        grads[fused] = eng.grad.clone()
        if fused == "1":
            for it in range(40):
                eng.forward_backward(rep)
                if it % 10 == 9:
                    torch.cuda.synchronize()
                    assert torch.equal(eng.grad, grads["1"]), it
            assert int(eng.xflags.sum()) == 0
            check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    scale = float(grads["0"].abs().max())
    assert scale > 0 and float((grads["0"] - grads["1"]).abs().max()) <= 2e-5 * scale


DROPOUT_CASES = [
    # cfg-1 shapes in latency mode (weights-through-LDS forward, matrix-core attention in the row slices, 4 backward slices)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, dropout=0.1), dict(batch=32, T=200, mask=-5, n_eps=40)),
    # one workgroup per sequence, VALU attention (head_dim 8)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, dropout=0.2), dict(batch=48, T=200, mask=-5, n_eps=60)),
    # matrix-core attention (head_dim 16), discrete tokens, D = 128 (register-direct stages)
    (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9, dropout=0.1), dict(batch=8, T=50, mask=8, n_eps=20)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, history_len=50, action_dim=8, pos="sin", dropout=0.3), dict(batch=6, T=200, mask=-5, n_eps=20, history=20)),
    # GRU gates and identity-reordered layers
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, gate="gru", dropout=0.1), dict(batch=16, T=200, mask=-5, n_eps=30)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, identity=True, dropout=0.1), dict(batch=8, T=200, mask=-5, n_eps=30)),
    # row-block tiled kernels: BASELINE config 4 / 5 shapes, a GRU-gated D = 128 net and an identity-reordered one
    (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=128, discrete=True, vocab_sizes=12, dropout=0.1), dict(batch=4, T=140, mask=11, n_eps=8)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, history_len=256, discrete=True, vocab_sizes=22, dropout=0.1), dict(batch=2, T=260, mask=21, n_eps=5)),
    (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9, gate="gru", dropout=0.2),
     dict(batch=8, T=50, mask=8, n_eps=20)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, history_len=50, identity=True, action_dim=8, dropout=0.1), dict(batch=8, T=200, mask=-5, n_eps=20)),
    # width-padded networks with dropout (round 5: keep masks keyed by (row, real column))
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, num_layers=2, history_len=50, dropout=0.1), dict(batch=16, T=200, mask=-5, n_eps=30)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=96, num_heads=4, num_layers=1, history_len=100, discrete=True, vocab_sizes=9, pos="sin", gate="gru", action_dim=4, dropout=0.2),
     dict(batch=4, T=120, mask=8, n_eps=10)),
]


@pytest.mark.parametrize("kw,run", DROPOUT_CASES)
def test_td_update_with_dropout(lib, kw, run, monkeypatch):
    """dropout > 0: keep masks of the embedding, the attention probabilities and the FFN output are a counter-based hash both
    the kernels and the oracle evaluate (the backward recomputes them); target forward in eval mode.  Same checks as without."""
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=run["batch"], T=run["T"], n_eps=run.get("n_eps", 9),
                                               mask=run["mask"], history=run.get("history"), device="cuda", test_lib=False)
    eng.td.dropout_seed = 4242
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=2)
    # the masks matter: the same update without dropout gives different Q-values for the train-mode passes, the same for the target
    q_drop = eng.q3.clone()
    if net.tiled and net.d_real > 0:       # a width-padded shape without dropout may run on the four-slice kernels (dtqn_limits.h, dtqn_ws_lite):
        monkeypatch.setenv("DTQN_WS_LITE_OFF", "1")       # keep the comparison inside one kernel family (bitwise-equal target rows)
    net0 = net_from_cfg(lib, O.NetCfg(**{**kw, "dropout": 0.0}))
    assert net0.tiled == net.tiled
    from dtqn_amd.learner import TdEngine
    eng0 = TdEngine(net0, run["batch"], history=eng.td.history)
    eng0.theta_pol.copy_(eng.theta_pol); eng0.theta_tgt.copy_(eng.theta_tgt)
    eng0.ep_idx.copy_(eng.ep_idx); eng0.start.copy_(eng.start)
    eng.forward_backward(rep); eng0.forward_backward(rep)
    torch.cuda.synchronize()
    n = run["batch"] * net.lp * net.ap
    qa, qb = eng.q3.reshape(3, -1), eng0.q3.reshape(3, -1)
    assert not torch.equal(qa[0], qb[0]) and not torch.equal(qa[1], qb[1]) and torch.equal(qa[2], qb[2])
