"""dtqn_actor_forward_batch -- what `run.py --num-envs N` and bench.py's vectorN_* lines run -- against the ORACLE directly
(it had only been compared with the single-actor entry point): ragged prefixes 1..L, N = 1, 5, 32, on both sides
of the two-workgroup switch, res and GRU gates, the tiled path; and its argument checks.  Emulation here, MI355X in
test_gpu_vector_parity.py."""
import ctypes

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O

from helpers import net_from_cfg, pack_theta, ptr


def run_batch(lib, net, theta, obs_list, act_list, device, stream=None, lens_override=None, n_max_override=None):
    """Pack N ragged prefixes the way VectorActor does and call the C entry point.  Returns (rc, q_last [N][A], q_all [N][n_max][A])."""
    L, Od, A, N = net.ctx_len, net.obs_dim, net.num_actions, len(obs_list)
    obs_bytes, act_bytes = N * L * Od * 4, (N * L + 3) & ~3
    total = obs_bytes + act_bytes + 4 * N
    cuda = device != "cpu"
    ctx_h = torch.zeros(total, dtype=torch.uint8)
    q_last = torch.full((N, A), float("nan"))
    if cuda:
        ctx_h, q_last = ctx_h.pin_memory(), q_last.pin_memory()
    buf = ctx_h.numpy()
    o = buf[:obs_bytes].view(np.float32).reshape(N, L, Od)
    a = buf[obs_bytes:obs_bytes + N * L].reshape(N, L)
    ln = buf[obs_bytes + act_bytes:].view(np.int32)
    n_max = 1
    for i, (ob, ac) in enumerate(zip(obs_list, act_list)):
        n = len(ob)
        o[i, :n], a[i, :n], ln[i] = ob, ac, n
        n_max = max(n_max, n)
    if lens_override is not None:
        ln[:] = lens_override
    if n_max_override is not None:
        n_max = n_max_override
    ctx_d = torch.zeros(total, dtype=torch.uint8, device=device)
    q_d = torch.full((N * L * A,), float("nan"), device=device)
    need = lib.dtqn_forward_workspace_floats(ctypes.byref(net), N)
    ws = torch.zeros(max(1, need), device=device)
    rc = lib.dtqn_actor_forward_batch(ctypes.byref(net), ptr(theta), ptr(ctx_h), ptr(ctx_d), N, n_max, ptr(q_d), ptr(q_last),
                                      ptr(ws) if need > 0 else None, 0, 0, 0, stream)
    if cuda:
        torch.cuda.synchronize()
    if rc == 0 and need > 0 and not net.tiled:
        assert not ws[lib.dtqn_td_xch_floats(ctypes.byref(net), N):].any()           # hand-over flags lowered again
    if rc != 0:
        return rc, None, None
    return rc, q_last.numpy().copy(), q_d.cpu().numpy()[:N * n_max * A].reshape(N, n_max, A)


CASES = [
    # cfg-1 network: N = 1 / 5 (latency mode: two workgroups per sequence once a prefix passes 32 rows) / 32 (one workgroup each)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50), (1, 5, 32)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=8, history_len=50, gate="gru", action_dim=8, pos="sin"), (5,)),
    (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9), (5,)),
    # row-block tiled path (context > 64)
    (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=128, discrete=True, vocab_sizes=12), (3,)),
]
EMU_CASES = [
    # (sizes the emulation finishes in seconds; the MI355X runs CASES)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50, num_layers=1), (2,)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, history_len=20, gate="gru", action_dim=4), (3,)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=4, num_layers=1, history_len=70, discrete=True, vocab_sizes=9, action_dim=8), (2,)),
    # a context of 10 on the 64-row instantiation of head_dim 16 (dtqn_limits.h), and one head of width 64 on the row-block path
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=4, history_len=10, num_layers=1), (3,)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=1, history_len=12, num_layers=1), (2,)),
    (dict(obs_dim=3, num_actions=3, inner_embed_size=48, num_heads=6, history_len=12, num_layers=1), (2,)),       # width-padded (DtqnNet.d_real)
]


def check_batched_actor_vs_oracle(lib, kw, sizes, device="cpu", stream=None, rounds=3):
    cfg = O.NetCfg(**kw)
    params = O.init_params(cfg, seed=41, perturb=True)
    net = net_from_cfg(lib, cfg)
    theta = torch.from_numpy(pack_theta(net, params)).to(device)
    L = cfg.history_len
    rng = np.random.default_rng(5)
    ot = torch.long if cfg.discrete else torch.float32
    worst = 0.0
    for N in sizes:
        for rnd in range(rounds):
            if rnd == 0:      # every length class at once: 1, the slice boundary, the full context
                lens = [1, L, max(1, L // 2), min(L, 33), min(L, 32)][:N] + [int(x) for x in rng.integers(1, L + 1, size=max(0, N - 5))]
            elif rnd == 1:    # all short: the 16- / 32-row instantiations of a 64-row context
                lens = [int(x) for x in rng.integers(1, min(L, 16) + 1, size=N)]
            else:
                lens = [int(x) for x in rng.integers(1, L + 1, size=N)]
            obs = [(rng.integers(0, cfg.vocab_sizes - 1, size=(n, cfg.obs_dim)) if cfg.discrete else rng.uniform(-1, 1, size=(n, cfg.obs_dim))).astype(np.float32)
                   for n in lens]
            act = [rng.integers(0, cfg.num_actions, size=n) for n in lens]
            rc, q_last, q_all = run_batch(lib, net, theta, obs, act, device, stream)
            assert rc == 0, (N, rnd, rc)
            for i, n in enumerate(lens):
                with torch.no_grad():
                    ref = O.forward(params, cfg, torch.as_tensor(obs[i][None], dtype=ot), torch.as_tensor(act[i][None, :, None], dtype=torch.long)).numpy()[0]
                scale = max(1.0, np.abs(ref).max())
                err = np.abs(q_last[i] - ref[-1]).max()
                worst = max(worst, err / scale)
                assert err <= 1e-4 * scale, (N, rnd, i, n, err)                          # north_star: Q within 1e-4
                assert np.abs(q_all[i, :n] - ref).max() <= 1e-4 * scale, (N, rnd, i, n)  # every live row, not only the reported one
                assert np.array_equal(q_last[i], q_all[i, n - 1])
    return worst


def check_batch_argument_errors(lib, device="cpu", stream=None):
    """Live-row counts outside 1..n_max, n_max outside 1..ctx_len and N < 1 are refused (DTQN_ERR_ARG) before anything is launched."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, history_len=50)
    net = net_from_cfg(lib, cfg)
    theta = torch.from_numpy(pack_theta(net, O.init_params(cfg, seed=1))).to(device)
    obs = [np.zeros((4, 3), np.float32), np.zeros((9, 3), np.float32)]
    act = [np.zeros(4, np.int64), np.zeros(9, np.int64)]
    ERR = B.DEFINES["DTQN_ERR_ARG"]
    assert run_batch(lib, net, theta, obs, act, device, stream)[0] == 0
    assert run_batch(lib, net, theta, obs, act, device, stream, lens_override=[0, 9])[0] == ERR
    assert run_batch(lib, net, theta, obs, act, device, stream, lens_override=[4, 10])[0] == ERR        # > n_max (9)
    assert run_batch(lib, net, theta, obs, act, device, stream, lens_override=[-3, 9])[0] == ERR
    assert run_batch(lib, net, theta, obs, act, device, stream, n_max_override=51)[0] == ERR
    assert run_batch(lib, net, theta, obs, act, device, stream, n_max_override=0)[0] == ERR


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


@pytest.mark.parametrize("kw,sizes", EMU_CASES)
def test_batched_actor_forward_vs_oracle_on_the_emulation(emu, kw, sizes):
    check_batched_actor_vs_oracle(emu, kw, sizes, rounds=1)


def test_batched_actor_argument_checks_on_the_emulation(emu):
    check_batch_argument_errors(emu)
