"""Object-surface checks on the CPU (kernels, where any run, on the test-only emulation): the CSV logger against rows
the reference's own logger wrote (G8), DTQN.reset_parameters against the statistics of the reference's init_weights
(G8), state_dict layout, checkpoint / load error handling."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from conftest import GOLDEN


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


@pytest.fixture(scope="module")
def g8():
    return np.load(os.path.join(GOLDEN, "G8_logging_init.npz"))


def test_csv_logger_writes_the_reference_rows(g8, tmp_path):
    """utils/logging_utils.py:42-109: same headers, column order, float formatting, append-on-resume."""
    from dtqn_amd.utils.logging_utils import CSVLogger
    envs = json.loads(str(g8["csv_envs"]))
    script = json.loads(str(g8["csv_script"]))
    path = str(tmp_path / "run")
    args = argparse.Namespace(envs=envs)
    lg = CSVLogger(path, args)
    for row, step in script[:2]:
        lg.log(row, step)
    lg2 = CSVLogger(path, args)                  # resume: no second header
    lg2.log(*script[2])
    assert open(path + "_results.csv", newline="").read() == str(g8["csv_results"])
    assert open(path + "_losses.csv", newline="").read() == str(g8["csv_losses"])


def _build(emu, cfg):
    from dtqn_amd.networks.dtqn import DTQN
    return DTQN(cfg.obs_dim, cfg.num_actions, cfg.embed_per_obs_dim, cfg.action_dim, cfg.inner_embed_size, cfg.num_heads,
                cfg.num_layers, cfg.history_len, dropout=0.0, gate=cfg.gate, identity=cfg.identity, pos=cfg.pos,
                discrete=cfg.discrete, vocab_sizes=cfg.vocab_sizes if cfg.discrete else None, bag_size=0, _test_lib=emu)


@pytest.mark.parametrize("name", ["default", "gru_a8_disc", "sin"])
def test_reset_parameters_has_the_reference_init_distribution(emu, g8, name):
    """utils/torch_utils.py:4-15 as applied by DTQN.__init__ (dtqn.py:156).  Tensors the reference initialises
    deterministically (biases, LayerNorm, learned positions = 0, masks, sinusoidal table) must be EQUAL in
    mean / min / max; N(0, 0.02) tensors must have the reference's moments within sampling error (5 sigma)."""
    st = json.loads(str(g8["init_stats"]))[name]
    cfg = O.NetCfg(**st["cfg"])
    torch.manual_seed(11)
    net = _build(emu, cfg)
    sd = net.state_dict()
    assert list(sd.keys()) == list(st["tensors"].keys())           # the reference's key order
    named = dict(net.named_parameters())
    for k, ref in st["tensors"].items():
        v = sd[k].double()
        assert list(v.shape) == ref["shape"], k
        fin = v[torch.isfinite(v)]
        if ref["requires_grad"] is not None and k in named:
            assert named[k].requires_grad == ref["requires_grad"], k
        if ref["std"] == 0.0 or k.endswith("attn_mask") or (k == "position_embedding.position_encoding" and cfg.pos != "learned"):
            for f, got in (("mean", fin.mean()), ("min", fin.min()), ("max", fin.max())):
                assert abs(float(got) - ref[f]) <= 1e-6, (k, f, float(got), ref[f])
        else:
            n = fin.numel()
            assert abs(ref["std"] - 0.02) < 5 * 0.02 / np.sqrt(2 * n) + 1e-4, (k, ref["std"])       # the fixture itself is N(0, 0.02)
            assert abs(float(fin.mean())) <= 5 * 0.02 / np.sqrt(n), (k, float(fin.mean()))
            assert abs(float(fin.std(unbiased=False)) - 0.02) <= 5 * 0.02 / np.sqrt(2 * n), (k, float(fin.std()))
            assert float(fin.abs().max()) <= 0.02 * 6.5, k
    if cfg.gate == "gru":
        # one gate instance shared by every layer (dtqn.py:107-131); w_z bias ends at 0 (init_weights runs after init_bias)
        a = sd["transformer_layers.0.attn_gate.w_r.weight"]
        b = sd["transformer_layers.1.attn_gate.w_r.weight"]
        assert a.data_ptr() == b.data_ptr() or torch.equal(a, b)
        assert float(sd["transformer_layers.0.mlp_gate.w_z.bias"].abs().max()) == 0.0
        n_unique = sum(p.numel() for p in net.parameters())
        n_ref = sum(int(np.prod(O.param_shapes(cfg)[k])) for k in O.trainable_keys(cfg)) + cfg.num_layers * cfg.history_len ** 2
        assert n_unique == n_ref                                    # parameters() de-duplicates the shared gates
    # two constructions draw different weights, the same seed draws the same
    torch.manual_seed(11)
    again = _build(emu, cfg)
    assert torch.equal(again.flat, net.flat)
    other = _build(emu, cfg)
    assert not torch.equal(other.flat, net.flat)


def test_load_state_dict_accepts_reference_keys_and_rejects_wrong_ones(emu):
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, num_layers=2, history_len=8, gate="gru", action_dim=4)
    net = _build(emu, cfg)
    params = O.init_params(cfg, seed=2, perturb=True)
    assert list(params.keys()) == list(net.state_dict().keys())
    net.load_state_dict({k: v.clone() for k, v in params.items()})          # strict
    from helpers import pack_theta
    assert np.array_equal(net.flat.numpy(), pack_theta(net.net, params))    # every view points into the flat buffer the kernels read
    bad = {k: v.clone() for k, v in params.items()}
    bad.pop("ffn.2.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)
    bad = {k: v.clone() for k, v in params.items()}
    bad["ffn.2.weight"] = torch.zeros(5, 16)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)


def test_unsupported_constructor_arguments_raise_like_documented(emu):
    from dtqn_amd.networks.dtqn import DTQN
    with pytest.raises(ValueError):
        DTQN(3, 3, 8, 0, 16, 2, 1, 8, pos=1, _test_lib=emu)                 # PosEnum(1) is rejected by the reference too
    with pytest.raises(ValueError):
        DTQN(3, 3, 8, 0, 16, 2, 1, 8, gate="lstm", _test_lib=emu)           # dtqn.py:114
