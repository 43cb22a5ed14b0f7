"""Shared test helpers: pack oracle parameter dicts into the flat theta layout, synthetic replay."""
import ctypes
import json

import numpy as np
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O


def net_from_cfg(lib, cfg: O.NetCfg):
    return B.make_net(lib, obs_dim=cfg.obs_dim, num_actions=cfg.num_actions, embed_per_obs_dim=cfg.embed_per_obs_dim,
                      action_dim=cfg.action_dim, inner_embed_size=cfg.inner_embed_size, num_heads=cfg.num_heads,
                      num_layers=cfg.num_layers, history_len=cfg.history_len, gate=cfg.gate, identity=cfg.identity,
                      pos=cfg.pos, discrete=cfg.discrete, vocab_sizes=cfg.vocab_sizes)


def pack_theta(net, params) -> np.ndarray:
    theta = np.zeros(net.n_theta, dtype=np.float32)
    for key, (off, shape) in B.param_table(net).items():
        v = params[key].detach().numpy().astype(np.float32).reshape(-1)
        assert v.size == int(np.prod(shape)), key
        theta[off:off + v.size] = v
    return theta


def unpack_flat(net, flat: np.ndarray, keys):
    tab = B.param_table(net)
    out = {}
    for k in keys:
        off, shape = tab[k]
        out[k] = flat[off:off + int(np.prod(shape))].reshape(shape).copy()
    return out


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(a.data_ptr())
