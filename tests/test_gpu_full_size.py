"""-m gpu: the BASELINE configurations at their FULL per-GPU batch (sizes at which the CPU oracle takes minutes per update),
checked through size-independent properties of the TD update:

  * batch independence: the Q-values (all three forwards) of a sequence do not depend on what else is in the batch -- the
    full-batch launch equals launches of 8-sequence subsets that the oracle-checked parity tests cover (different grid sizes,
    weight-gradient kernels and latency-mode decisions on the two sides);
  * linearity of the gradient in the batch: the mean-loss gradient of the full batch is the average of the gradients of its
    equal parts (each part scaled by its share), and the loss statistic likewise;
  * determinism: two runs of the full-size update from the same state are bit-identical.
"""
import numpy as np
import pytest
import torch

from oracle import dtqn_oracle as O

from helpers import net_from_cfg, pack_theta

pytestmark = pytest.mark.gpu

FULL = {
    "cfg2": (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50), 256, 200, -5),
    "cfg3": (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9), 512, 50, 8),
    # the same on the whole-sequence kernels (DTQN_TRAIN_TILED=0): the library's policy trains cfg 3 on the row-block twin of the net
    "cfg3_whole_sequence": (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9), 512, 50, 8),
    "cfg4": (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2, history_len=128, discrete=True, vocab_sizes=12), 128, 250, 11),
    "cfg5": (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, num_layers=2, history_len=256, discrete=True, vocab_sizes=22), 32, 256, 21),
}


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


def _engine(lib, cfg, batch, T, mask, params, replay_arrays):
    from dtqn_amd.learner import DeviceReplay, TdEngine
    net = net_from_cfg(lib, cfg)
    eng = TdEngine(net, batch)
    eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, params[0])))
    eng.theta_tgt.copy_(torch.from_numpy(pack_theta(net, params[1])))
    obs, act, rew, done, lens = replay_arrays
    rep = DeviceReplay(obs.shape[0], T, cfg.obs_dim, mask, eng.device)
    rep.obs.copy_(torch.from_numpy(obs)); rep.actions.copy_(torch.from_numpy(act)); rep.rewards.copy_(torch.from_numpy(rew))
    rep.dones.copy_(torch.from_numpy(done)); rep.ep_len.copy_(torch.from_numpy(lens))
    return net, eng, rep


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_update_properties(lib, name, monkeypatch):
    if name.endswith("_whole_sequence"):
        monkeypatch.setenv("DTQN_TRAIN_TILED", "0")
    kw, Bn, T, mask = FULL[name]
    cfg = O.NetCfg(**kw)
    L, A = cfg.history_len, cfg.num_actions
    rng = np.random.Generator(np.random.PCG64(17))
    E = Bn + 40
    lens = rng.integers(5, T + 1, size=E).astype(np.int32)
    if cfg.discrete:
        obs = rng.integers(0, cfg.vocab_sizes - 1, size=(E, T + 1, cfg.obs_dim)).astype(np.float32)   # tokens below the mask value (cfg 3: SURVEY.md section 8, V = 9, mask 8)
    else:
        obs = rng.uniform(-1, 1, size=(E, T + 1, cfg.obs_dim)).astype(np.float32)
    obs[np.arange(T + 1)[None, :] > lens[:, None]] = mask
    act = rng.integers(0, A, size=(E, T + 1)).astype(np.uint8)
    rew = rng.choice(np.array([0, 0, 0, 1, -1], dtype=np.float32), size=(E, T))
    done = np.ones((E, T), dtype=np.uint8)
    live = np.arange(T)[None, :] < lens[:, None]
    done[live] = 0
    done[np.arange(E), lens - 1] = 1
    rew[~live] = 0
    arrays = (obs, act, rew, done, lens)
    params = (O.init_params(cfg, seed=5, perturb=True), O.init_params(cfg, seed=6, perturb=True))
    eps = rng.permutation(E)[:Bn].astype(np.int32)
    starts = np.array([rng.integers(0, max(0, lens[e] - L) + 1) for e in eps], dtype=np.int32)

    net, eng, rep = _engine(lib, cfg, Bn, T, mask, params, arrays)
    eng.set_indices(eps, starts)
    eng.forward_backward(rep)
    torch.cuda.synchronize()
    q_full = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :A].copy()
    g_full = eng.grad.cpu().numpy().copy()
    sp = eng.stats_partial.cpu().numpy().reshape(-1, 8)
    assert np.isfinite(q_full).all() and np.isfinite(g_full).all()
    # determinism at full size
    eng.forward_backward(rep)
    torch.cuda.synchronize()
    assert np.array_equal(eng.grad.cpu().numpy(), g_full)

    # --- the same sequences in parts: Q per sequence, gradient and loss as averages of the parts
    parts = 4
    pb = Bn // parts
    g_sum = np.zeros_like(g_full)
    netp, engp, repp = _engine(lib, cfg, pb, T, mask, params, arrays)
    scale = max(1.0, float(np.abs(q_full).max()))
    for p in range(parts):
        sl = slice(p * pb, (p + 1) * pb)
        engp.set_indices(eps[sl], starts[sl])
        engp.forward_backward(repp)
        torch.cuda.synchronize()
        q_p = engp.q3.cpu().numpy().reshape(3, pb, net.lp, net.ap)[:, :, :L, :A]
        assert np.abs(q_p - q_full[:, sl]).max() <= 1e-5 * scale, (name, p, np.abs(q_p - q_full[:, sl]).max())
        g_sum += engp.grad.cpu().numpy()
    g_avg = g_sum / parts
    gmax = np.abs(g_full).max()
    assert gmax > 0
    assert np.abs(g_avg - g_full).max() <= 2e-5 * gmax, (name, np.abs(g_avg - g_full).max() / gmax)

    # --- 8-sequence subsets, the size the oracle-checked tests run: Q of every 16th group equals the full-batch rows
    net8, eng8, rep8 = _engine(lib, cfg, 8, T, mask, params, arrays)
    for start in range(0, Bn, max(8, Bn // 4)):
        sl = slice(start, start + 8)
        eng8.set_indices(eps[sl], starts[sl])
        eng8.forward_backward(rep8)
        torch.cuda.synchronize()
        q_8 = eng8.q3.cpu().numpy().reshape(3, 8, net.lp, net.ap)[:, :, :L, :A]
        # (cfg 3: the 8-sequence engine runs the whole-sequence kernels in latency mode, the full batch the row-block ones --
        #  two kernel families, two summation orders)
        same_family = eng8.net.tiled == eng.net.tiled
        assert np.abs(q_8 - q_full[:, sl]).max() <= (1e-5 if same_family else 5e-5) * scale, (name, start)
    # and those 8 sequences against the CPU oracle (one subset: the oracle needs seconds per forward at these widths)
    sl = slice(0, 8)
    ot = torch.long if cfg.discrete else torch.float32
    rows = starts[sl, None] + np.arange(L)[None, :]
    o = torch.as_tensor(obs[eps[sl, None], rows], dtype=ot)
    a = torch.as_tensor(act[eps[sl, None], rows][..., None].astype(np.int64))
    with torch.no_grad():
        ref = O.forward(params[0], cfg, o, a).numpy()
    assert np.abs(q_full[0, sl] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
