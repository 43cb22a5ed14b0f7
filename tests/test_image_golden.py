"""Image observation embedding (five 3x3 convolutions + Linear, dtqn/networks/representations.py:77-130) against
tests/golden/G11_image.npz -- outputs of THE REFERENCE's DTQN built with obs_dim = (C, H, W) (tests/golden/make_golden.py gen_G11):
the oracle on the CPU, then the engine (dtqn_img_encode / dtqn_img_backward around the row-block update) on the test-only HIP
emulation here and on the MI355X in test_gpu_image.py."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from helpers import flat_from_params, pack_theta

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["a", "b"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, "G11_image.npz"), allow_pickle=False)
    kw = json.loads(str(z[f"{name}_cfg"]))
    kw["image"] = tuple(kw["image"])
    cfg = O.NetCfg(**kw)
    seed = int(z[f"{name}_seed"])
    pol = O.init_params(cfg, seed=seed, perturb=True)
    tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
    cs = O.param_checksum(pol)
    assert np.isfinite(cs) and cs > 0
    assert cs == pytest.approx(float(z[f"{name}_pol_checksum"]), rel=1e-12)
    return z, cfg, pol, tgt


def golden_batch(z, name, cfg):
    L = cfg.history_len
    rows, acts = z[f"{name}_rows"], z[f"{name}_actions"]
    return O.Batch(obss=torch.as_tensor(rows[:, :L]), actions=torch.as_tensor(acts[:, :L], dtype=torch.long),
                   rewards=torch.as_tensor(z[f"{name}_rewards"], dtype=torch.float32), next_obss=torch.as_tensor(rows[:, 1:]),
                   next_actions=torch.as_tensor(acts[:, 1:], dtype=torch.long), dones=torch.as_tensor(z[f"{name}_dones"], dtype=torch.long))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_image_embedding_matches_the_reference(name):
    z, cfg, pol, tgt = load_case(name)
    batch = golden_batch(z, name, cfg)
    grads, out = O.td_gradients(pol, tgt, cfg, batch, 0.99, cfg.history_len)
    for w, key in ((4, "q_all"), (5, "q_next_pol"), (6, "q_next_tgt")):
        ref = z[f"{name}_{key}"]
        assert np.abs(out[w].detach().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), key
    assert float(out[0].detach()) == pytest.approx(float(z[f"{name}_loss"]), rel=1e-5)
    flat = np.concatenate([grads[k].numpy().ravel() for k in O.trainable_keys(cfg)])
    ref = z[f"{name}_grad_flat"]
    assert flat.shape == ref.shape and np.abs(flat - ref).max() <= 1e-4 * np.abs(ref).max()


def test_oracle_forward_at_the_minihack_pixel_crop_size():
    z = np.load(os.path.join(GOLDEN, "G11_image.npz"), allow_pickle=False)
    kw = json.loads(str(z["fwd144_cfg"])); kw["image"] = tuple(kw["image"])
    cfg = O.NetCfg(**kw)
    pol = O.init_params(cfg, seed=int(z["fwd144_seed"]), perturb=True)
    with torch.no_grad():
        q = O.forward(pol, cfg, torch.as_tensor(z["fwd144_obs"])).numpy()
    assert O.image_feat(cfg) == 128 * 18 * 18
    assert np.abs(q - z["fwd144_q"]).max() <= 5e-6 * max(1.0, np.abs(z["fwd144_q"]).max())


# --------------------------------------------------------------------------- engine
def image_net(lib, cfg):
    return B.make_net(lib, obs_dim=1, image=cfg.image, num_actions=cfg.num_actions, inner_embed_size=cfg.inner_embed_size,
                      num_heads=cfg.num_heads, num_layers=cfg.num_layers, history_len=cfg.history_len, gate=cfg.gate, identity=cfg.identity,
                      pos=cfg.pos, dropout=cfg.dropout)


def check_engine_vs_g11(lib, name, device="cpu", test_lib=True):
    """TdEngine on the fixture's pixel windows: Q of the three forwards within north_star's 1e-4, the full gradient (convolutions,
    embedding linear, transformer, head) within 2e-4 of max|g|, the loss statistic, and a finite optimizer step."""
    from dtqn_amd.learner import DeviceReplay, TdEngine
    z, cfg, pol, tgt = load_case(name)
    net = image_net(lib, cfg)
    assert net.tiled == 1 and net.img_feat == O.image_feat(cfg)
    Bn, L, A = int(z[f"{name}_B"]), cfg.history_len, cfg.num_actions
    eng = TdEngine(net, Bn, lr=3e-4, gamma=0.99, history=L, tuf=10_000, _test_lib=lib if test_lib else None, device=None if test_lib else device)
    net = eng.net
    eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, pol)))
    eng.theta_tgt.copy_(torch.from_numpy(pack_theta(net, tgt)))
    rep = DeviceReplay(Bn, L, cfg.image, 0, eng.device)
    rep.obs.copy_(torch.from_numpy(z[f"{name}_rows"].reshape(Bn, L + 1, -1)))
    rep.actions.copy_(torch.from_numpy(z[f"{name}_actions"][:, :, 0].astype(np.uint8)))
    rep.rewards.copy_(torch.from_numpy(z[f"{name}_rewards"][:, :, 0]))
    rep.dones.copy_(torch.from_numpy(z[f"{name}_dones"][:, :, 0].astype(np.uint8)))
    rep.ep_len.fill_(L)
    eng.set_indices(np.arange(Bn, dtype=np.int32), np.zeros(Bn, dtype=np.int32))
    eng.forward_backward(rep)
    q3 = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :A]
    for w, key in enumerate(("q_all", "q_next_pol", "q_next_tgt")):
        ref = z[f"{name}_{key}"]
        assert np.abs(q3[w] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (name, key, np.abs(q3[w] - ref).max())
    keys = O.trainable_keys(cfg)
    ref_flat, off = {}, 0
    for k in keys:
        n = int(np.prod(pol[k].shape))
        ref_flat[k] = torch.from_numpy(z[f"{name}_grad_flat"][off:off + n].reshape(tuple(pol[k].shape)).copy())
        off += n
    ref_g = flat_from_params(net, ref_flat, keys)
    got = eng.grad.cpu().numpy()
    gmax = np.abs(ref_g).max()
    tab = B.param_table(net)
    worst = {k: float(np.abs(got[tab[k][0]:tab[k][0] + int(np.prod(tab[k][1]))] - ref_g[tab[k][0]:tab[k][0] + int(np.prod(tab[k][1]))]).max()) for k in keys}
    bad = {k: v / gmax for k, v in worst.items() if v > 2e-4 * gmax}
    assert not bad, (name, bad)
    eng.clip_adam()
    st = eng.read_stats()
    assert st["nonfinite"] == 0.0 and st["step"] == 1
    assert abs(st["td_error"] - float(z[f"{name}_loss"])) <= 2e-4 * max(1.0, float(z[f"{name}_loss"]))
    assert bool(torch.isfinite(eng.theta_pol).all())


def check_module_forward_vs_g11(lib, device="cpu", tags=("a", "fwd144")):
    """DTQN(obs_dim = (C, H, W)).forward on [B, n, C, H, W] pixels: the reference's constructor surface, state_dict keys
    (obs_embedding.observation_embedding.{0,2,4,6,8,11}.*) and Q-values, incl. the 3 x 144 x 144 MiniHack crop."""
    from dtqn_amd.networks.dtqn import DTQN
    z = np.load(os.path.join(GOLDEN, "G11_image.npz"), allow_pickle=False)
    for tag in tags:
        kw = json.loads(str(z[f"{tag}_cfg"])); kw["image"] = tuple(kw["image"])
        cfg = O.NetCfg(**kw)
        pol = O.init_params(cfg, seed=int(z[f"{tag}_seed"]), perturb=True)
        m = DTQN(cfg.image, cfg.num_actions, cfg.embed_per_obs_dim, 0, cfg.inner_embed_size, cfg.num_heads, cfg.num_layers, cfg.history_len,
                 pos=cfg.pos, **({"_test_lib": lib} if lib is not None else {}))
        m._allow_cpu = lib is not None
        m = m.to(device)
        assert [k for k in m.state_dict() if k.startswith("obs_embedding")] == [k for k in O.state_dict_keys(cfg) if k.startswith("obs_embedding")]
        m.load_state_dict({k: (pol[k] if k in pol else v) for k, v in m.state_dict().items()})
        if tag == "a":
            obs, ref = z["a_rows"][:, :cfg.history_len], z["a_q_all"]
        else:
            obs, ref = z["fwd144_obs"], z["fwd144_q"]
        q = m(torch.as_tensor(obs)).cpu().numpy()
        assert np.abs(q - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (tag, np.abs(q - ref).max())
        with pytest.raises(AssertionError):
            m(torch.zeros(1, 2, 3, 8, 8, dtype=torch.uint8))           # dtqn.py:176-179: "Obs dim is incorrect"


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


@pytest.mark.parametrize("name", NAMES)
def test_engine_on_the_emulation_matches_the_reference_with_image_observations(emu, name):
    check_engine_vs_g11(emu, name)


def test_module_forward_on_the_emulation(emu):
    check_module_forward_vs_g11(emu, tags=("a",))          # (the 144 x 144 crop runs on the MI355X: test_gpu_image.py)


def test_image_net_coverage_limits(emu):
    """What the gfx950 kernels do not cover is refused at construction (NotImplementedError), like every other variant."""
    with pytest.raises(NotImplementedError):
        B.make_net(emu, obs_dim=1, image=(3, 16, 16), num_actions=4, inner_embed_size=64, num_heads=4, history_len=6, action_dim=16)
    with pytest.raises(NotImplementedError):
        B.make_net(emu, obs_dim=1, image=(4, 16, 16), num_actions=4, inner_embed_size=64, num_heads=4, history_len=6)
    net = B.make_net(emu, obs_dim=1, image=(3, 144, 144), num_actions=8, inner_embed_size=64, num_heads=8, history_len=50)
    assert (net.img_h1, net.img_h3, net.img_h5, net.img_feat, net.img_k1) == (72, 36, 18, 128 * 18 * 18, 32)
