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
                      pos=cfg.pos, discrete=cfg.discrete, vocab_sizes=cfg.vocab_sizes, dropout=cfg.dropout,
                      bag_size=cfg.bag_size)


def pack_theta(net, params) -> np.ndarray:
    """Oracle / reference-shaped tensors -> the engine's flat buffer (width-padded networks: zeros in the padding, DtqnNet.d_real)."""
    theta = np.zeros(net.n_theta, dtype=np.float32)
    for key, (off, shape) in B.param_table(net).items():
        v = B.pad_param(net, key, params[key].detach().numpy().astype(np.float32), tuple(shape)).reshape(-1)
        assert v.size == int(np.prod(shape)), key
        theta[off:off + v.size] = v
    return theta


def unpack_flat(net, flat: np.ndarray, keys):
    tab = B.param_table(net)
    real = B.param_table(net, net.d_real) if net.d_real else tab
    out = {}
    for k in keys:
        off, shape = tab[k]
        out[k] = np.ascontiguousarray(B.unpad_param(net, k, flat[off:off + int(np.prod(shape))].reshape(shape), real[k][1])).copy()
    return out


def padding_mask(net) -> np.ndarray:
    """True at the trainable entries of the flat buffer that are PADDING of a width-padded network (all False otherwise)."""
    m = np.zeros(net.n_trainable, dtype=bool)
    if net.d_real:
        real = B.param_table(net, net.d_real)
        for k, (off, shape) in B.param_table(net).items():
            if off < net.n_trainable:
                n = int(np.prod(shape))
                m[off:off + n] = B.pad_param(net, k, np.ones(real[k][1], dtype=np.float32), tuple(shape)).reshape(-1) == 0
    return m


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(a.data_ptr())


# --------------------------------------------------------------------------- #
# TD-update harness shared by the CPU-emulation tests and the -m gpu tests
# --------------------------------------------------------------------------- #
def fill_device_replay(dev_replay, host_buf):
    """Copy an oracle ReplayOracle's arrays into a dtqn_amd DeviceReplay (test set-up only)."""
    dev_replay.obs.copy_(torch.from_numpy(host_buf.obss))
    dev_replay.actions.copy_(torch.from_numpy(host_buf.actions[:, :, 0]))
    dev_replay.rewards.copy_(torch.from_numpy(host_buf.rewards[:, :, 0]))
    dev_replay.dones.copy_(torch.from_numpy(host_buf.dones[:, :, 0].astype(np.uint8)))
    dev_replay.ep_len.copy_(torch.from_numpy(host_buf.episode_lengths.astype(np.int32)))


def oracle_batch(host_buf, eps, starts, discrete):
    o, a, r, no, na, d, _ = host_buf.gather(eps, starts)
    ot = torch.long if discrete else torch.float32
    return O.Batch(obss=torch.as_tensor(o, dtype=ot), actions=torch.as_tensor(a, dtype=torch.long),
                   rewards=torch.as_tensor(r, dtype=torch.float32), next_obss=torch.as_tensor(no, dtype=ot),
                   next_actions=torch.as_tensor(na, dtype=torch.long), dones=torch.as_tensor(d, dtype=torch.long))


def flat_from_params(net, params, keys):
    """Concatenate oracle tensors into the engine's flat layout (trainable region only)."""
    tab = B.param_table(net)
    flat = np.zeros(net.n_trainable, dtype=np.float32)
    for k in keys:
        off, shape = tab[k]
        v = B.pad_param(net, k, params[k].detach().numpy(), tuple(shape)).reshape(-1)
        flat[off:off + v.size] = v
    return flat


def make_td_case(lib, cfg, *, seed, batch, T, n_eps, mask, history=None, tuf=10_000, lr=3e-4, device="cpu", test_lib=True,
                 weight_scale=1.0):
    """Oracle learner + dtqn_amd TdEngine on identical parameters and an identical synthetic replay.
    weight_scale: factor on every weight MATRIX of the perturbed parameter set (0.1: std 0.2 -> 0.02, the init_weights scale of
    utils/torch_utils.py:4-15, with biases / LayerNorm / positions still moved off their init values: |Q| < 1, so the Q tolerance of
    check_td_updates is the ABSOLUTE 1e-4 of north_star there)."""
    import random as pyrandom
    from dtqn_amd.learner import DeviceReplay, TdEngine
    from oracle.replay_oracle import ReplayOracle, synth_fill
    net = net_from_cfg(lib, cfg)
    pol = O.init_params(cfg, seed=seed, perturb=True)
    tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
    if weight_scale != 1.0:
        for params in (pol, tgt):
            for k, v in params.items():
                if v.dim() >= 2 and not k.endswith("attn_mask") and k != "position_embedding.position_encoding":
                    v.mul_(weight_scale)
    hist = cfg.history_len if history is None else history
    oracle = O.OracleLearner(cfg, pol, lr=lr, gamma=0.99, history=hist, tuf=tuf, target=tgt)
    host = ReplayOracle((n_eps + 2) * T, cfg.obs_dim, mask, T, cfg.history_len)
    synth_fill(host, np.random.Generator(np.random.PCG64(seed + 1000)), n_eps, cfg.discrete, cfg.vocab_sizes, cfg.num_actions)
    eng = TdEngine(net, batch, lr=lr, gamma=0.99, history=hist, tuf=tuf, _test_lib=lib if test_lib else None,
                   device=None if test_lib else device)
    eng.theta_pol.copy_(torch.from_numpy(pack_theta(net, pol)))
    eng.theta_tgt.copy_(torch.from_numpy(pack_theta(net, tgt)))
    rep = DeviceReplay(host.max_size, T, cfg.obs_dim, mask, eng.device)
    fill_device_replay(rep, host)
    pyrandom.seed(seed)
    return net, oracle, host, eng, rep


def engine_probe(cfg, net, eng):
    """The engine's own ReLU activation patterns (decoded from its saved ballots / hidden record) and double-DQN argmax
    choices, in the order oracle.td_gradients consumes them: gradients are compared CONDITIONAL on these."""
    Bn, L, A = eng.batch, cfg.history_len, cfg.num_actions
    q3 = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :A].copy()
    act = eng.act.cpu().numpy()[:Bn * net.act_stride].reshape(Bn, net.act_stride)
    D = cfg.inner_embed_size
    fld = lambda off, w: torch.from_numpy(act[:, off:off + net.lp * w].reshape(Bn, net.lp, w)[:, :L].copy() > 0)

    def ballots(off, w):
        """Decode the engine's ReLU ballot words (include/dtqn_hip.h, al_m1/al_mh/al_m2) into [B, L, w] booleans."""
        ct = w // 16
        nwords = (net.lp // 16) * ct * 4
        words = np.ascontiguousarray(act[:, off:off + 2 * nwords]).view(np.uint64).reshape(Bn, nwords)
        rows, cols = np.meshgrid(np.arange(L), np.arange(w), indexing="ij")
        widx = ((rows >> 4) * ct + (cols >> 4)) * 4 + (rows & 3)
        bit = (((rows >> 2) & 3) << 4) + (cols & 15)
        return torch.from_numpy(((words[:, widx] >> bit.astype(np.uint64)) & np.uint64(1)).astype(bool))
    masks = []
    Dp = net.d_model                  # the records are d_model wide; a width-padded network's real columns are the first D of them
    for l in range(cfg.num_layers):
        base = net.ao_layer0 + l * net.act_layer_stride
        m_h = ballots(base + net.al_mh, 4 * Dp)
        assert torch.equal(m_h, fld(base + net.al_h, 4 * Dp)), "hidden ballot disagrees with the saved hidden"
        m1, m2 = ballots(base + net.al_m1, Dp), ballots(base + net.al_m2, Dp)
        assert not m_h[..., 4 * D:].any() and not m1[..., D:].any() and not m2[..., D:].any(), "a padded unit is active"
        masks += [m1[..., :D], m_h[..., :4 * D], m2[..., :D]]
    hh = fld(net.ao_hh, Dp)
    assert not hh[..., D:].any()
    masks.append(hh[..., :D])
    return {"masks": masks, "argmax": torch.from_numpy(q3[1].argmax(-1))}


def check_td_updates(cfg, net, oracle, host, eng, rep, n_updates, q_tol=1e-4, grad_rtol=2e-4, one_call=False, q_rel=False,
                     pipelined=False, draw_seed=1234, report_as=None):
    """Run n_updates on both sides from identical (episode, start) draws and compare every stage:
    the three Q tensors, pre-clip gradients, statistics, parameters after the step.
    one_call: the whole update through dtqn_td_update, as the agent's train() issues it, instead of stage by stage;
    everything compared is still left behind by that call.
    pipelined: the update as DtqnAgent.train() issues it with the device sampler (dtqn_amd/learner.py: the window draw inside the
    kernels, policy passes as four row slices, the next update's target pass inside this update's backward launch -- or on the side
    stream for row-block nets): the engine draws its own windows, the oracle batch is built from the (episode, start) pairs the
    kernels left in ep_idx / start, everything else is compared as in the other modes.  The caller has enabled the pipeline.
    report_as: name under which the worst absolute Q error / gradient error of the run go to gpurun_out/parity_report.json."""
    keys = O.trainable_keys(cfg)
    Bn, L, A = eng.batch, cfg.history_len, cfg.num_actions
    net = eng.net                  # the net the update runs on: the caller's, or its row-block twin (dtqn_td_prefers_tiled)
    worst = {"q_abs_err": 0.0, "q_abs_max": 0.0, "grad_err_over_max": 0.0, "updates": n_updates}
    for it in range(n_updates):
        if pipelined:
            assert getattr(eng, "_pipe", None) is not None and cfg.bag_size == 0 and cfg.dropout == 0
            n_valid, exclude = min(host.pos[0], host.max_size), host.pos[0] % host.max_size
            eng.sample_in_forward(n_valid, exclude, draw_seed)
            pre = eng.theta_pol.cpu().numpy()[:net.n_trainable].copy()
            eng.update(rep)        # dtqn_td_update_pipelined (ride) / staged launches with the side stream (row-block nets)
            idx = eng._idx_dev.cpu().numpy()
            eps, starts = idx[0].astype(np.int64), idx[1].astype(np.int64)
            # the draw is the reference's distribution (replay_buffer.py:141-158): finished slots minus the one in progress,
            # start on {0 .. max(0, len - L)}
            assert ((eps >= 0) & (eps < n_valid) & (eps != exclude)).all(), eps
            assert ((starts >= 0) & (starts <= np.maximum(0, host.episode_lengths[eps] - L))).all(), (eps, starts)
            batch = oracle_batch(host, eps, starts, cfg.discrete)
        else:
            eps, starts = host.sample_indices(Bn)
            batch = oracle_batch(host, eps, starts, cfg.discrete)
            eng.set_indices(eps, starts)
        if cfg.bag_size > 0:       # a synthetic bag per window (the engine takes whatever the sampler hands it)
            rng = np.random.Generator(np.random.PCG64(77 + it))
            bo = rng.integers(0, cfg.vocab_sizes, (Bn, cfg.bag_size, cfg.obs_dim)).astype(np.float32) if cfg.discrete \
                else rng.random((Bn, cfg.bag_size, cfg.obs_dim), dtype=np.float32)
            ba = rng.integers(0, cfg.num_actions, (Bn, cfg.bag_size, 1))
            eng.set_bag(bo, ba)
            batch.bag_obss = torch.as_tensor(bo, dtype=torch.long if cfg.discrete else torch.float32)
            batch.bag_actions = torch.as_tensor(ba, dtype=torch.long)
        if pipelined:
            pass                   # the whole update ran above
        else:
            pre = eng.theta_pol.cpu().numpy()[:net.n_trainable].copy()
            if one_call:
                eng.update(rep)
            else:
                eng.forward_backward(rep)
        # --- Q-values of the three forwards
        # The gradient is discontinuous at every ReLU kink and at ties of the double-DQN argmax; two
        # correct fp32 implementations can sit on different sides of a kink whose pre-activation is
        # ~1e-7 (observed), which moves dW by percents.  So gradients are compared CONDITIONAL on the
        # engine's own activation pattern (read from its saved activations) and argmax choices, and
        # the number / size of the disagreements is bounded separately.
        q3 = eng.q3.cpu().numpy().reshape(3, Bn, net.lp, net.ap)[:, :, :L, :A].copy()
        probe = engine_probe(cfg, net, eng)
        D = cfg.inner_embed_size
        # dropout: the engine keys its keep masks by (dropout_seed, optimizer steps so far); the oracle evaluates the same hash
        drop = O.DropSpec(cfg.dropout, int(eng.td.dropout_seed), it) if cfg.dropout > 0 else None
        grads, out = O.td_gradients(oracle.pol, oracle.tgt, cfg, batch, oracle.gamma, oracle.history, probe, drop)
        assert not probe["masks"], "oracle consumed fewer ReLU masks than the engine saved"
        n_relu = Bn * L * (6 * D * cfg.num_layers + D)
        assert probe.get("relu_flips", 0) <= max(2, n_relu // 20000), probe
        assert probe.get("max_flip_preact", 0.0) <= 2e-5 * max(1.0, float(out[4].detach().abs().max())), probe
        assert probe.get("argmax_flips", 0) <= max(1, Bn * L // 500), probe
        assert probe.get("max_flip_qgap", 0.0) <= 2e-4 * max(1.0, float(out[4].detach().abs().max())), probe
        for w, ref in enumerate((out[4], out[5], out[6])):
            err = np.abs(q3[w] - ref.detach().numpy()).max()
            # north_star's 1e-4 is ABSOLUTE at the scale a DTQN run lives at (|Q| of order 1: rewards in [-1, 1]; init-scale and
            # trained parameters are pinned absolutely by test_gpu_parity_holes' init-scale cases, G1 and the G12 loop).  These
            # synthetic cases use weights of std 0.2 -- ten times the init scale -- to make every term count; |Q| reaches 1e1 - 1e2
            # there and two correct fp32 summation orders differ by a few 1e-5 * |Q|max, so the bound scales with |Q|max beyond 1.
            qmax = float(ref.detach().abs().max())
            worst["q_abs_err"], worst["q_abs_max"] = max(worst["q_abs_err"], float(err)), max(worst["q_abs_max"], qmax)
            assert err <= q_tol * max(1.0, qmax), (it, w, err, qmax)
        # --- gradients (flat, engine layout), relative to the largest entry as in the oracle's own golden check
        ref_flat = flat_from_params(net, grads, keys)
        got = eng.grad.cpu().numpy().copy()
        gerr = np.abs(got - ref_flat).max()
        worst["grad_err_over_max"] = max(worst["grad_err_over_max"], float(gerr / np.abs(ref_flat).max()))
        assert gerr <= grad_rtol * np.abs(ref_flat).max(), (it, gerr, np.abs(ref_flat).max(), probe)
        pad = padding_mask(net)
        if pad.any():                  # width-padded network: the padding takes no gradient at all
            assert not got[:net.n_trainable][pad].any(), (it, int(np.count_nonzero(got[:net.n_trainable][pad])))
        # --- optimizer step + statistics
        if not one_call and not pipelined:
            eng.clip_adam()
        st = eng.read_stats()
        # teacher-force the oracle from the engine's pre-step parameters?  No: both sides started
        # identical and took identical steps so far; compare the step itself on solid elements.
        ref_stats = oracle.update(batch, grads_and_out=(grads, out))
        assert st["nonfinite"] == 0.0
        assert st["step"] == it + 1
        for k in ("td_error", "grad_norm", "qvalue_max", "qvalue_mean", "qvalue_min", "target_max", "target_mean", "target_min"):
            assert st[k] == ref_stats[k] or abs(st[k] - ref_stats[k]) <= 2e-4 * max(1.0, abs(ref_stats[k])), (it, k, st[k], ref_stats[k])
        post = eng.theta_pol.cpu().numpy()[:net.n_trainable].copy()
        assert not post[pad].any()     # ... and stays zero through the optimizer step
        ref_post = flat_from_params(net, oracle.pol, keys)
        d = np.abs(post - ref_post)
        solid = np.abs(ref_flat) >= 1e-3 * np.abs(ref_flat).max()
        if it == 0:
            assert d[solid].max() <= 2e-6, (it, d[solid].max())
        # |Adam step| <= lr at k = 1 and <= lr*(1-b1)/sqrt(1-b2) = 3.17 lr in general
        # (+ one ulp of the largest parameter: the step is rounded into theta)
        assert np.abs(post - pre).max() <= (1.001 if it == 0 else 3.2) * oracle.lr + float(np.spacing(np.abs(pre).max()))
        assert d.max() <= 2.002 * oracle.lr * (it + 1)
        # keep the two sides in lock-step for the next iteration (removes chaotic drift from the comparison)
        eng.theta_pol[:net.n_trainable].copy_(torch.from_numpy(ref_post))
        tab = B.param_table(net)
        for k in keys:
            off, shape = tab[k]
            n = int(np.prod(shape))
            eng.adam_m[off:off + n].copy_(B.pad_param(net, k, oracle.opt.m[k], tuple(shape)).reshape(-1))
            eng.adam_v[off:off + n].copy_(B.pad_param(net, k, oracle.opt.v[k], tuple(shape)).reshape(-1))
        # target sync parity
        tgt_ref = pack_theta(net, oracle.tgt)[:net.n_trainable]
        tgt_got = eng.theta_tgt.cpu().numpy()[:net.n_trainable].copy()
        if (it + 1) % oracle.tuf == 0:
            assert st["target_synced"] == 1.0
            assert np.abs(tgt_got - post).max() == 0.0
            eng.theta_tgt[:net.n_trainable].copy_(torch.from_numpy(tgt_ref))
        else:
            assert st["target_synced"] == 0.0
            assert np.abs(tgt_got - tgt_ref).max() == 0.0
    if pipelined:
        worst["pipeline"] = {"used": int(eng._pipe["used"]), "inline": int(eng._pipe["inline"]), "ride": bool(eng._pipe["ride"])}
    if report_as is not None:
        parity_report(report_as, worst)
    return worst


def parity_report(name: str, payload: dict) -> None:
    """Measured error margins of a parity check -> gpurun_out/parity_report.json (merged back from the GPU box; committed per round
    under profiles/)."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, "parity_report.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    data[name] = payload
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
