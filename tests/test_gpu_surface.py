"""-m gpu: the reference's OBJECT surface on the device: `DTQN(...).load_state_dict(reference-keyed
dict)` -> `forward` against the reference's own Q-values (G1, G3), checkpoint save -> load -> identical next update,
`--save-policy` state_dict round trip, the asynchronous statistics the CSV rows are made of."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import dtqn_oracle as O

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _module(cfg):
    from dtqn_amd.networks.dtqn import DTQN
    return DTQN(cfg.obs_dim, cfg.num_actions, cfg.embed_per_obs_dim, cfg.action_dim, cfg.inner_embed_size, cfg.num_heads,
                cfg.num_layers, cfg.history_len, dropout=0.0, gate=cfg.gate, identity=cfg.identity, pos=cfg.pos,
                discrete=cfg.discrete, vocab_sizes=cfg.vocab_sizes if cfg.discrete else None, bag_size=0).to("cuda")


def _q(net, cfg, obss, actions):
    o = torch.as_tensor(obss, dtype=torch.long if cfg.discrete else torch.float32)
    a = torch.as_tensor(actions, dtype=torch.long)
    return net(o, a).cpu().numpy()              # the reference call: net(obss, actions) -> [B, seq, A]


def test_module_forward_with_reference_state_dict_G1():
    """dtqn/networks/dtqn.py:41-59,158-164: construct, load the reference-keyed state_dict (incl. the frozen attn_mask
    entries), call forward with the reference's argument types; outputs = what the reference network produced."""
    z = np.load(os.path.join(GOLDEN, "G1_cfg1_td.npz"))
    cfg = O.NetCfg(**json.loads(str(z["cfg"])))
    seed = int(z["seed"])
    pol, tgt = _module(cfg), _module(cfg)
    assert list(pol.state_dict().keys()) == O.state_dict_keys(cfg)
    pol.load_state_dict({k: v.clone() for k, v in O.init_params(cfg, seed=seed, perturb=True).items()})
    tgt.load_state_dict({k: v.clone() for k, v in O.init_params(cfg, seed=seed + 1, perturb=True).items()})
    tgt.eval()
    scale = 1.0          # absolute tolerance (north_star: 1e-4 fp32)
    assert np.abs(_q(pol, cfg, z["batch0_obss"], z["batch0_actions"]) - z["q_all"]).max() <= 1e-4 * scale
    assert np.abs(_q(pol, cfg, z["batch0_next_obss"], z["batch0_next_actions"]) - z["q_next_pol"]).max() <= 1e-4 * scale
    assert np.abs(_q(tgt, cfg, z["batch0_next_obss"], z["batch0_next_actions"]) - z["q_next_tgt"]).max() <= 1e-4 * scale
    # state_dict -> torch.save -> load -> load_state_dict (run.py --save-policy, :337-340,463-466)
    sd = {k: v.cpu() for k, v in pol.state_dict().items()}
    fresh = _module(cfg)
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.flat, pol.flat)
    with pytest.raises(AssertionError):
        pol(torch.zeros(1, 51, 3), torch.zeros(1, 51, 1, dtype=torch.long))       # dtqn.py:170-173
    with pytest.raises(AssertionError):
        pol(torch.zeros(1, 5, 4), torch.zeros(1, 5, 1, dtype=torch.long))         # dtqn.py:175-179


def test_module_forward_with_reference_state_dict_G3():
    """cfg 3 (whole-sequence kernels, discrete tokens as int64 like the reference passes them), cfg 4 / 5 (tiled path)."""
    z = np.load(os.path.join(GOLDEN, "G3_cfg345_td.npz"))
    for name in json.loads(str(z["names"])):
        p = name + "/"
        cfg = O.NetCfg(**json.loads(str(z[p + "cfg"])))
        seed = int(z[p + "seed"])
        pol = _module(cfg)
        pol.load_state_dict({k: v.clone() for k, v in O.init_params(cfg, seed=seed, perturb=True).items()})
        scale = max(1.0, np.abs(z[p + "q_all"]).max())      # std-0.2 stress weights: see tests/helpers.py
        err = np.abs(_q(pol, cfg, z[p + "batch0_obss"], z[p + "batch0_actions"]) - z[p + "q_all"]).max()
        assert err <= 1e-4 * scale, (name, err)
        err = np.abs(_q(pol, cfg, z[p + "batch0_next_obss"], z[p + "batch0_next_actions"]) - z[p + "q_next_pol"]).max()
        assert err <= 1e-4 * scale, (name, err)


def _agent(env, seed, **kw):
    from dtqn_amd.utils.agent_utils import get_agent
    from dtqn_amd.utils.random import set_global_seed
    set_global_seed(seed, env)
    return get_agent("DTQN", [env], 8, 0, 64, 8_000, torch.device("cuda"), 3e-4, 32, 50, -1, 50, 4, 0.99, 8, 2, 0.0,
                     False, kw.pop("gate", "res"), "learned", 0, **kw)


@pytest.mark.parametrize("gate", ["res", "gru"])
def test_checkpoint_round_trip_on_device(tmp_path, gate):
    """dqn.py:212-327 / run.py:345-352,482-488: save_checkpoint -> a fresh agent -> load_checkpoint -> both take the
    same next updates bit for bit (parameters, Adam moments, statistics), including across a hard target sync."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import LinearAnneal
    from dtqn_amd.utils.logging_utils import RunningAverage
    env = envs.make("DiscreteCarFlag-v0")
    a = _agent(env, 2, gate=gate)
    runpy.prepopulate(a, 7000, [env])
    for _ in range(5):
        a.train()
    eps = LinearAnneal(1.0, 0.1, 10)
    eps.anneal()
    ras = [RunningAverage(10) for _ in range(3)]
    ras[0].add(0.5); ras[1].add(-1.0); ras[2].add(37)
    path = str(tmp_path / "ck")
    a.save_checkpoint(path, "wid", ras[0], ras[1], ras[2], eps)
    b = _agent(envs.make("DiscreteCarFlag-v0"), 99, gate=gate)
    wid, s, r, l, ev = b.load_checkpoint(path)
    assert wid == "wid" and ev == eps.val and s.mean() == 0.5 and r.mean() == -1.0 and l.mean() == 37 and b.num_train_steps == 5
    assert torch.equal(a.policy_network.flat, b.policy_network.flat) and torch.equal(a.target_network.flat, b.target_network.flat)
    assert torch.equal(a.engine.adam_m, b.engine.adam_m) and torch.equal(a.engine.adam_v, b.engine.adam_v)
    for x, y in zip((a.replay_buffer.dev.obs, a.replay_buffer.dev.actions, a.replay_buffer.dev.rewards, a.replay_buffer.dev.dones, a.replay_buffer.dev.ep_len),
                    (b.replay_buffer.dev.obs, b.replay_buffer.dev.actions, b.replay_buffer.dev.rewards, b.replay_buffer.dev.dones, b.replay_buffer.dev.ep_len)):
        assert torch.equal(x, y)
    assert a.td_errors.mean() == b.td_errors.mean() and list(a.grad_norms.q) == list(b.grad_norms.q)
    assert b.load_mini_checkpoint(path)["step"] == 5
    for it in range(4):                       # tuf = 4: update 8 syncs the target on both
        st = random.getstate()
        a.train()
        random.setstate(st)                   # `random` is process-global: give b the same window draw
        b.train()
        torch.cuda.synchronize()
        assert torch.equal(a.policy_network.flat, b.policy_network.flat), it
        assert torch.equal(a.target_network.flat, b.target_network.flat), it
        assert torch.equal(a.engine.adam_v, b.engine.adam_v), it
    assert a.td_errors.mean() == b.td_errors.mean() and len(b.td_errors.q) == 9


def test_loading_a_checkpoint_into_an_agent_that_already_trained(tmp_path):
    """The statistics ring restarts cleanly: host call counters, device call counter and ring tags agree
    after load_checkpoint on a used agent."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.logging_utils import RunningAverage
    env = envs.make("DiscreteCarFlag-v0")
    a = _agent(env, 3)
    runpy.prepopulate(a, 7000, [env])
    for _ in range(3):
        a.train()
    path = str(tmp_path / "ck")
    ras = [RunningAverage(10) for _ in range(3)]
    a.save_checkpoint(path, None, ras[0], ras[1], ras[2], Constant(0.1))
    for _ in range(7):
        a.train()
    a.load_checkpoint(path)                    # the same, already used, agent
    assert a.num_train_steps == 3 and len(a.td_errors.q) == 3
    for _ in range(5):
        a.train()
    m = a.td_errors.mean()
    assert np.isfinite(m) and len(a.td_errors.q) == 8 and a.num_train_steps == 8
    assert int(a.engine.step_counter[1].item()) == 8


@pytest.mark.parametrize("heads,padded", [(6, (64, 8, 8)), (4, (64, 4, 16))])
def test_module_at_a_padded_width_speaks_the_reference_shapes(heads, padded):
    """DTQN(inner_embed_size=48, num_heads=6) on the device (DtqnNet.d_real: runs at width 64 with two all-zero heads): reference-shaped
    state_dict in and out, forward == the oracle at width 48 on full contexts and prefixes, policy -> target copy."""
    from helpers import padding_mask
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=heads, num_layers=2, history_len=50, pos="sin")
    m = _module(cfg)
    assert (m.net.d_real, m.net.d_model, m.net.num_heads, m.net.head_dim) == (48,) + padded
    params = O.init_params(cfg, seed=11, perturb=True)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in params.items()}
    m.load_state_dict({k: v.clone() for k, v in params.items()})
    pad = torch.from_numpy(padding_mask(m.net)).cuda()
    assert pad.any() and not m.flat[:m.net.n_trainable][pad].any()
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), params[k]), k
    rng = np.random.default_rng(3)
    for n in (1, 17, 50):
        obs = rng.uniform(-1, 1, (4, n, 3)).astype(np.float32)
        act = np.zeros((4, n, 1), dtype=np.int64)
        with torch.no_grad():
            ref = O.forward(params, cfg, torch.as_tensor(obs), torch.as_tensor(act)).numpy()
        assert np.abs(_q(m, cfg, obs, act) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), n
    tgt = _module(cfg)
    tgt.load_state_dict(m.state_dict())
    assert torch.equal(tgt.flat, m.flat)
