"""-m gpu: dropout on the MI355X against the reference's own train() (tests/golden/G10_dropout.npz, see test_dropout_golden.py), and the
keep rate / scaling / eval-mode properties of the kernels' masks measured on the device."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import dtqn_oracle as O

from helpers import make_td_case, net_from_cfg, pack_theta, ptr
from test_dropout_golden import NAMES, check_engine_vs_g10

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


@pytest.mark.parametrize("name", NAMES)
def test_engine_on_the_gpu_matches_the_reference_with_dropout(lib, name):
    """res: cfg-1 shapes (latency mode, weights-through-LDS forward, four backward slices); gru / ident: whole-sequence kernels at D = 32;
    bag: row-block tiled kernels."""
    check_engine_vs_g10(lib, name, device="cuda", test_lib=False)


@pytest.mark.parametrize("split", ["default", "0"])
def test_dropout_keep_rate_scaling_and_eval_mode_on_the_device(lib, split, monkeypatch):
    """nn.Dropout's definition measured on what the forward kernel wrote: x0 = dropout(embedding + position) in the activation record
    is 0 for a fraction p of the elements and the undropped value x 1 / (1 - p) for the rest, the kept set is exactly the hash the
    oracle evaluates, the masks of two optimizer steps are independent, and an eval-mode forward ignores the dropout setting."""
    if split == "0":
        monkeypatch.setenv("DTQN_ROW_SPLIT", "0")
    p = 0.3
    kw = dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50)
    cfgp, cfg0 = O.NetCfg(dropout=p, **kw), O.NetCfg(**kw)
    Bn, L, D = 32, 50, 64
    recs = {}
    for tag, cfg in (("p", cfgp), ("0", cfg0)):
        net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=5, batch=Bn, T=120, n_eps=40, mask=-5, device="cuda", test_lib=False)
        eng.td.dropout_seed = 99
        eps, starts = np.arange(Bn, dtype=np.int32), np.zeros(Bn, dtype=np.int32)
        eng.set_indices(eps, starts)
        xs = []
        for step in range(2):
            eng.step_counter[1] = step
            eng.forward_backward(rep)
            act = eng.act.cpu().numpy()[:Bn * eng.net.act_stride].reshape(Bn, eng.net.act_stride)
            xs.append(act[:, eng.net.ao_x0:eng.net.ao_x0 + eng.net.lp * D].reshape(Bn, eng.net.lp, D)[:, :L].copy())
        recs[tag] = (xs, eng.q3.cpu().numpy().reshape(3, -1).copy())
    (xp0, xp1), q_p = recs["p"]
    (x00, _), q_0 = recs["0"]
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    keep0, keep1 = xp0 != 0, xp1 != 0
    n = keep0.size
    assert abs(keep0.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n)                 # keep rate
    assert np.abs(xp0[keep0] - x00[keep0] * scale).max() <= 2e-6 * np.abs(x00).max() # survivors scaled by 1 / (1 - p)
    idx = (np.arange(L)[:, None] * D + np.arange(D)[None, :]).astype(np.uint64)
    for b in (0, 7, 31):
        want = O.drop_keep(O.DropSpec(p, 99, 0, 0), b, O.DROP_EMB, 0, idx)
        assert np.array_equal(keep0[b] | (x00[b] == 0), want | (x00[b] == 0)), b     # the oracle's hash, element for element
    agree = (keep0 == keep1).mean()                                                  # independent masks at the next optimizer step
    assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.01
    assert not np.array_equal(q_p[0], q_0[0]) and not np.array_equal(q_p[1], q_0[1]) and np.array_equal(q_p[2], q_0[2])   # target: eval mode
    # dtqn_forward (inference) is an eval-mode forward whatever the net's dropout says
    params = O.init_params(cfg0, seed=3, perturb=True)
    obs = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, size=(2, L, 3)).astype(np.float32)).cuda()
    act = torch.zeros(2, L, dtype=torch.uint8, device="cuda")
    outs = []
    for cfg in (cfg0, cfgp):
        net = net_from_cfg(lib, cfg)
        theta = torch.from_numpy(pack_theta(net, params)).cuda()
        q = torch.full((2, L, 3), float("nan"), device="cuda")
        assert lib.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), 2, L, ptr(q), None) == 0
        torch.cuda.synchronize()
        outs.append(q.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
