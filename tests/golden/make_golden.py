#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  It imports the
reference's modules unmodified, with two stub modules (`gym`, `wandb`; plus `pyglet`
for the env files) placed in sys.modules because those third-party packages are not
installed and are not on the arithmetic path.  One harness patch
is applied: ReplayBuffer.episode_lengths is cast to int64 after construction to restore
the pinned numpy-1.22 semantics of `uint8 - int` (replay_buffer.py:69,152; SURVEY.md
section 4 quirk 1).

Only DATA is written: inputs, expected outputs, seeds and version stamps (.npz / .json).
No reference source, bytecode or pickled module travels.  Weights are produced by
oracle.dtqn_oracle.init_params (numpy PCG64, key order = state_dict order) and loaded
into the reference network with load_state_dict, so both sides regenerate them from the
seed; a checksum of the weights is stored to catch generator drift.

Usage:  python tests/golden/make_golden.py            (writes next to this file)
"""
from __future__ import annotations

import importlib.util
import json
import os
import random
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


# --------------------------------------------------------------------------- #
# stubs for absent third-party modules
# --------------------------------------------------------------------------- #
def install_stubs():
    gym = types.ModuleType("gym")

    class Env:
        def seed(self, seed=None):
            return [seed]

    class _Space:
        def seed(self, seed=None):
            return [seed]

    class Discrete(_Space):
        def __init__(self, n):
            self.n = n

    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec)
            self.shape = self.nvec.shape

    class MultiBinary(_Space):
        def __init__(self, n):
            self.n = n

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete, spaces.MultiDiscrete, spaces.MultiBinary, spaces.Box = Discrete, MultiDiscrete, MultiBinary, Box
    gym.Env, gym.Wrapper, gym.spaces = Env, Wrapper, spaces
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    utils.seeding = seeding
    gym.utils = utils
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding})
    sys.modules["wandb"] = types.ModuleType("wandb")
    pyglet = types.ModuleType("pyglet")
    canvas = types.ModuleType("pyglet.canvas")
    xlib = types.ModuleType("pyglet.canvas.xlib")

    class NoSuchDisplayException(Exception):
        pass

    xlib.NoSuchDisplayException = NoSuchDisplayException
    sys.modules.update({"pyglet": pyglet, "pyglet.canvas": canvas, "pyglet.canvas.xlib": xlib})


install_stubs()
sys.path.insert(0, REF)

from dtqn.networks.dtqn import DTQN as RefDTQN                      # noqa: E402
from dtqn.agents.dtqn import DtqnAgent as RefAgent                  # noqa: E402
from dtqn.buffers.replay_buffer import ReplayBuffer as RefBuffer    # noqa: E402
from utils.context import Context as RefContext                     # noqa: E402
from utils.epsilon_anneal import LinearAnneal as RefAnneal          # noqa: E402
from utils.logging_utils import RunningAverage as RefRunAvg         # noqa: E402
import utils.random as ref_random                                   # noqa: E402

from oracle import dtqn_oracle as O                                  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


STAMP = {"torch": torch.__version__, "numpy": np.__version__,
         "reference": "kevslinger/DTQN @ 2024_08_07", "ref_pins": "torch==1.11.0 numpy==1.22.4"}


def checksum(params):
    return O.param_checksum(params)          # finite entries only (the causal masks are -inf)


def make_ref_net(cfg: O.NetCfg, params):
    net = RefDTQN(tuple(cfg.image) if cfg.image is not None else cfg.obs_dim, cfg.num_actions, cfg.embed_per_obs_dim, cfg.action_dim, cfg.inner_embed_size,
                  cfg.num_heads, cfg.num_layers, cfg.history_len, dropout=cfg.dropout, gate=cfg.gate,
                  identity=cfg.identity, pos=cfg.pos, discrete=cfg.discrete,
                  vocab_sizes=cfg.vocab_sizes if cfg.discrete else None, bag_size=cfg.bag_size)
    assert list(net.state_dict().keys()) == O.state_dict_keys(cfg), "state_dict key order drifted"
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == O.param_shapes(cfg)[k], k
    net.load_state_dict({k: v.clone() for k, v in params.items()})
    return net


def synth_episodes(rng, n_eps, T, cfg: O.NetCfg, min_len=3):
    """Synthetic replay content in the shape of SURVEY.md section 8d."""
    eps = []
    for _ in range(n_eps):
        n = int(rng.integers(min_len, T + 1))
        if cfg.discrete:
            obs = rng.integers(0, cfg.vocab_sizes - 1, size=(n + 1, cfg.obs_dim)).astype(np.int64)
        else:
            obs = rng.uniform(-1, 1, size=(n + 1, cfg.obs_dim)).astype(np.float32)
        act = rng.integers(0, cfg.num_actions, size=n)
        rew = rng.choice(np.array([0, 0, 0, 1, -1], dtype=np.float32), size=n)
        done = np.zeros(n, dtype=bool)
        done[-1] = True
        eps.append((obs, act, rew, done))
    return eps


def make_ref_agent(cfg: O.NetCfg, pol_params, tgt_params, B, T, buf_eps, mask, lr=3e-4, gamma=0.99,
                   history=None, tuf=10_000):
    history = cfg.history_len if history is None else history
    it = iter([pol_params, tgt_params])
    agent = RefAgent(lambda: make_ref_net(cfg, next(it)), buffer_size=buf_eps * T, device=torch.device("cpu"),
                     env_obs_length=cfg.obs_dim, max_env_steps=T, obs_mask=mask, num_actions=cfg.num_actions,
                     is_discrete_env=cfg.discrete, learning_rate=lr, batch_size=B, context_len=cfg.history_len,
                     gamma=gamma, history=history, target_update_frequency=tuf, bag_size=cfg.bag_size)
    # DqnAgent.__init__ hard-copies policy -> target (dqn.py:49); restore the distinct target weights
    agent.target_network.load_state_dict({k: v.clone() for k, v in tgt_params.items()})
    agent.replay_buffer.episode_lengths = agent.replay_buffer.episode_lengths.astype(np.int64)  # quirk 1
    return agent


def fill_agent(agent, episodes):
    for obs, act, rew, done in episodes:
        agent.context_reset(obs[0])
        for t in range(len(act)):
            agent.observe(obs[t + 1], int(act[t]), float(rew[t]), bool(done[t]))
        agent.replay_buffer.flush()


def run_ref_updates(agent, n_updates):
    """Run agent.train() n_updates times, capturing the sampled batch, pre-clip grads and stats."""
    rec = {"batches": [], "grads": [], "norms": [], "stats": [], "pre": [], "post": [], "m": [], "v": []}
    tparams = [p for p in agent.policy_network.parameters() if p.requires_grad]
    flat = lambda ts: np.concatenate([t.detach().numpy().ravel() for t in ts])
    orig_sample, orig_sample_bag = agent.replay_buffer.sample, agent.replay_buffer.sample_with_bag

    def sample(bs):
        out = orig_sample(bs)
        rec["batches"].append([np.array(a) for a in out])
        return out

    def sample_with_bag(bs, bag):
        out = orig_sample_bag(bs, bag)
        rec["batches"].append([np.array(a) for a in out])
        return out

    agent.replay_buffer.sample = sample
    agent.replay_buffer.sample_with_bag = sample_with_bag
    orig_clip = torch.nn.utils.clip_grad_norm_

    def clip(params, max_norm, **kw):
        params = list(params)
        rec["grads"].append([None if p.grad is None else p.grad.detach().clone() for p in params])
        n = orig_clip(params, max_norm, **kw)
        rec["norms"].append(float(n))
        return n

    torch.nn.utils.clip_grad_norm_ = clip
    try:
        for _ in range(n_updates):
            rec["pre"].append(flat(tparams))
            agent.train()
            rec["post"].append(flat(tparams))
            rec["m"].append(flat([agent.optimizer.state[p]["exp_avg"] for p in tparams]))
            rec["v"].append(flat([agent.optimizer.state[p]["exp_avg_sq"] for p in tparams]))
            rec["stats"].append({
                "td_error": agent.td_errors.q[-1], "grad_norm": agent.grad_norms.q[-1],
                "qvalue_max": agent.qvalue_max.q[-1], "qvalue_mean": agent.qvalue_mean.q[-1],
                "qvalue_min": agent.qvalue_min.q[-1], "target_max": agent.target_max.q[-1],
                "target_mean": agent.target_mean.q[-1], "target_min": agent.target_min.q[-1]})
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
        agent.replay_buffer.sample = orig_sample
        agent.replay_buffer.sample_with_bag = orig_sample_bag
    return rec


def batch_to_npz(prefix, b):
    names = ["obss", "actions", "rewards", "next_obss", "next_actions", "dones", "ep_lens"]
    return {f"{prefix}{n}": np.asarray(a) for n, a in zip(names, b)}


def ref_q(net, obss, actions, discrete):
    with torch.no_grad():
        o = torch.as_tensor(obss, dtype=torch.long if discrete else torch.float32)
        a = torch.as_tensor(actions, dtype=torch.long)
        return net(o, a).numpy()


def td_case(cfg: O.NetCfg, seed, B, T, n_eps, mask, n_updates, store_grads=True, store_params=True,
            lr=3e-4, history=None, tuf=10_000, trace=False):
    """One learner fixture: reference agent runs n_updates TD updates on a synthetic buffer."""
    random.seed(seed)
    ref_random.RNG.rng = np.random.Generator(np.random.PCG64(seed))
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    pol = O.init_params(cfg, seed=seed, perturb=True)
    tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
    agent = make_ref_agent(cfg, pol, tgt, B, T, n_eps + 2, mask, lr=lr, history=history, tuf=tuf)
    fill_agent(agent, synth_episodes(rng, n_eps, T, cfg))
    # Q-values of the three forwards on the first batch, BEFORE any update
    state = random.getstate()
    first = [np.array(a) for a in agent.replay_buffer.sample(B)]
    random.setstate(state)
    agent.eval_off()
    out = {"cfg": json.dumps(cfg.to_json()), "seed": seed, "B": B, "T": T, "mask": mask,
           "lr": lr, "gamma": 0.99, "history": cfg.history_len if history is None else history, "tuf": tuf,
           "n_updates": n_updates, "pol_checksum": checksum(pol), "tgt_checksum": checksum(tgt),
           "stamp": json.dumps(STAMP)}
    out["q_all"] = ref_q(agent.policy_network, first[0], first[1], cfg.discrete)
    out["q_next_pol"] = ref_q(agent.policy_network, first[3], first[4], cfg.discrete)
    out["q_next_tgt"] = ref_q(agent.target_network, first[3], first[4], cfg.discrete)
    rec = run_ref_updates(agent, n_updates)
    assert all(np.array_equal(a, b) for a, b in zip(first, rec["batches"][0]))
    for i, b in enumerate(rec["batches"]):
        out.update(batch_to_npz(f"batch{i}_", b))
    keys = O.trainable_keys(cfg)
    named = dict(agent.policy_network.named_parameters())          # de-duplicated, registration order
    pnames = [n for n, p in agent.policy_network.named_parameters()]
    out["stats"] = json.dumps(rec["stats"])
    out["grad_norms"] = np.array(rec["norms"], dtype=np.float64)
    if store_grads:
        # pre-clip gradients of update 0, concatenated in oracle.trainable_keys order
        gl = {n: g for n, g in zip(pnames, rec["grads"][0]) if g is not None}
        assert sorted(gl) == sorted(keys)
        out["grad0_flat"] = np.concatenate([gl[k].numpy().ravel() for k in keys])
    sd = agent.policy_network.state_dict()
    if store_params:
        out["final_flat"] = np.concatenate([sd[k].numpy().ravel() for k in keys])
    out["final_checksum"] = checksum({k: sd[k] for k in keys})
    # parameters after the FIRST update (exact pin of Adam step k=1 and of the clip coefficient)
    if store_grads and store_params:
        out["post0_flat"] = rec["post"][0]
    # teacher-forcing trace: state before/after every update, so each Adam step can be checked from
    # the reference's own pre-state (free-running comparisons are chaotic, see tests/test_oracle_golden.py)
    if trace:
        for i in range(n_updates):
            out[f"pre{i}_flat"], out[f"post{i}_flat"] = rec["pre"][i], rec["post"][i]
            out[f"m{i}_flat"], out[f"v{i}_flat"] = rec["m"][i], rec["v"][i]
            if i > 0:
                out[f"grad{i}_flat"] = np.concatenate([g.numpy().ravel() for g in rec["grads"][i] if g is not None])
    out["q_all_final"] = ref_q(agent.policy_network, first[0], first[1], cfg.discrete)
    return out


# --------------------------------------------------------------------------- #
# fixture groups
# --------------------------------------------------------------------------- #
def gen_G1():
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    out = td_case(cfg, seed=11, B=32, T=200, n_eps=60, mask=-5, n_updates=3)
    np.savez_compressed(os.path.join(HERE, "G1_cfg1_td.npz"), **out)


def variant_cfgs():
    """24 cases: the full gate x identity x pos factorial (12), each with two of the six
    (a_embed, obs kind) pairs in rotation, so every pair is seen under four network variants."""
    obs_kinds = [dict(obs_dim=3, discrete=False, vocab_sizes=0, mask=-5),
                 dict(obs_dim=10, discrete=True, vocab_sizes=9, mask=8),
                 dict(obs_dim=1, discrete=True, vocab_sizes=22, mask=21)]
    pairs = [(a, ok) for a in (0, 4) for ok in obs_kinds]
    cases = []
    combo = 0
    for gate in ("res", "gru"):
        for identity in (False, True):
            for pos in ("learned", "sin", "none"):
                for j in range(2):
                    a_embed, ok = pairs[(2 * combo + j + combo // 3) % 6]
                    cfg = O.NetCfg(obs_dim=ok["obs_dim"], num_actions=4, embed_per_obs_dim=8,
                                   action_dim=a_embed, inner_embed_size=16, num_heads=2, num_layers=2,
                                   history_len=8, gate=gate, identity=identity, pos=pos,
                                   discrete=ok["discrete"], vocab_sizes=ok["vocab_sizes"])
                    cases.append((cfg, ok["mask"]))
                combo += 1
    return cases


def gen_G2():
    out = {}
    names = []
    for idx, (cfg, mask) in enumerate(variant_cfgs()):
        name = f"v{idx:02d}"
        names.append(name)
        # history < context on every third case pins the [:, -history:] slice (dtqn.py:240-241)
        hist = 5 if idx % 3 == 0 else None
        c = td_case(cfg, seed=100 + idx, B=2, T=12, n_eps=6, mask=mask, n_updates=3, history=hist, tuf=2,
                    store_params=(idx % 4 == 0), trace=(idx in (0, 13, 22)))
        for k, v in c.items():
            out[f"{name}/{k}"] = v
    out["names"] = json.dumps(names)
    np.savez_compressed(os.path.join(HERE, "G2_variants_td.npz"), **out)


def gen_G3():
    """cfg 3/4/5 shapes, outputs + stats only, B=2."""
    cfgs = {
        "cfg3": (O.NetCfg(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2,
                          history_len=50, discrete=True, vocab_sizes=9), 8, 50),
        "cfg4": (O.NetCfg(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2,
                          history_len=128, discrete=True, vocab_sizes=12), 11, 250),
        "cfg5": (O.NetCfg(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, num_layers=2,
                          history_len=256, discrete=True, vocab_sizes=22), 21, 256),
    }
    out = {}
    for name, (cfg, mask, T) in cfgs.items():
        c = td_case(cfg, seed=31, B=2, T=T, n_eps=5, mask=mask, n_updates=1, store_grads=False,
                    store_params=False)
        for k, v in c.items():
            out[f"{name}/{k}"] = v
    out["names"] = json.dumps(list(cfgs))
    np.savez_compressed(os.path.join(HERE, "G3_cfg345_td.npz"), **out)


def gen_G4():
    """Variable-length actor forward (dtqn/agents/dtqn.py:81-107): Q[:, -1] at seq len 1,2,17,50."""
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    cfg_a = O.NetCfg(obs_dim=3, num_actions=3, action_dim=8, inner_embed_size=64, num_heads=8, num_layers=2,
                     history_len=50, gate="gru", pos="sin")
    out = {"stamp": json.dumps(STAMP)}
    for tag, c in (("res", cfg), ("gru_a8_sin", cfg_a)):
        params = O.init_params(c, seed=41, perturb=True)
        net = make_ref_net(c, params)
        net.train()
        rng = np.random.Generator(np.random.PCG64(41))
        out[f"{tag}/cfg"] = json.dumps(c.to_json())
        out[f"{tag}/checksum"] = checksum(params)
        for n in (1, 2, 17, 50):
            obs = rng.uniform(-1, 1, size=(1, n, 3)).astype(np.float32)
            act = rng.integers(0, 3, size=(1, n, 1))
            out[f"{tag}/n{n}_obs"] = obs
            out[f"{tag}/n{n}_act"] = act
            out[f"{tag}/n{n}_q"] = ref_q(net, obs, act, False)
    np.savez_compressed(os.path.join(HERE, "G4_actor_varlen.npz"), **out)


def gen_G5():
    """ReplayBuffer: scripted store/flush sequence (with wrap-around and an in-progress episode)
    -> full array dump + sample() under random.seed (replay_buffer.py:71-168)."""
    out = {"stamp": json.dumps(STAMP)}
    for tag, (O_len, mask, T, L, dtype) in {"cont": (3, -5, 12, 5, np.float32), "disc": (2, 7, 9, 4, np.int64)}.items():
        buf = RefBuffer(buffer_size=5 * T, env_obs_length=O_len, obs_mask=mask, max_episode_steps=T, context_len=L)
        buf.episode_lengths = buf.episode_lengths.astype(np.int64)
        rng = np.random.Generator(np.random.PCG64(51))
        script = []
        for ep in range(8):                                     # 8 episodes into 5 slots -> wraps
            n = int(rng.integers(2, T + 1)) if ep != 3 else T   # one full-length episode
            o0 = (rng.uniform(-1, 1, O_len).astype(np.float32) if dtype == np.float32
                  else rng.integers(0, mask, O_len))
            buf.store_obs(o0)
            script.append(("store_obs", np.asarray(o0, dtype=np.float64)))
            steps = n if ep != 7 else 3                          # episode 7 stays in progress (no flush)
            for t in range(steps):
                o = (rng.uniform(-1, 1, O_len).astype(np.float32) if dtype == np.float32
                     else rng.integers(0, mask, O_len))
                a, r, d = int(rng.integers(0, 4)), float(rng.choice([0.0, 1.0, -1.0])), bool(t == n - 1)
                buf.store(o, a, r, d, t + 1)
                script.append(("store", np.concatenate([np.asarray(o, dtype=np.float64), [a, r, d, t + 1]])))
            if ep != 7:
                buf.flush()
                script.append(("flush", np.zeros(0)))
        out[f"{tag}/script_ops"] = json.dumps([s[0] for s in script])
        out[f"{tag}/script_args"] = np.array([np.pad(s[1], (0, O_len + 4 - len(s[1]))) for s in script])
        out[f"{tag}/meta"] = json.dumps(dict(obs_len=O_len, mask=mask, T=T, L=L, buffer_size=5 * T,
                                             discrete=dtype != np.float32))
        out[f"{tag}/obss"], out[f"{tag}/actions"] = buf.obss.copy(), buf.actions.copy()
        out[f"{tag}/rewards"], out[f"{tag}/dones"] = buf.rewards.copy(), buf.dones.copy()
        out[f"{tag}/episode_lengths"] = buf.episode_lengths.copy()
        out[f"{tag}/pos"] = np.array(buf.pos)
        out[f"{tag}/can_sample_4"] = buf.can_sample(4)
        out[f"{tag}/can_sample_7"] = buf.can_sample(7)
        out[f"{tag}/can_sample_8"] = buf.can_sample(8)
        random.seed(77)
        for i in range(3):
            s = buf.sample(6)
            out.update(batch_to_npz(f"{tag}/sample{i}_", s))
    np.savez_compressed(os.path.join(HERE, "G5_replay.npz"), **out)


def gen_G6():
    """Env traces: CarFlag / Memory (obs, reward, done, info.is_success) for fixed seeds and action
    scripts (envs/car_flag.py:70-159, envs/memory_cards.py:64-116), without the TimeLimit wrapper
    (gym 0.18's TimeLimit is third-party; the build restates its step-count semantics)."""
    car = load_by_path("ref_car_flag", os.path.join(REF, "envs/car_flag.py"))
    mem = load_by_path("ref_memory", os.path.join(REF, "envs/memory_cards.py"))
    out = {"stamp": json.dumps(STAMP)}
    for seed in (1, 7):
        env = car.CarFlag(discrete=True)
        env.seed(seed)
        rng = np.random.Generator(np.random.PCG64(seed + 5))
        obs_l, rew_l, done_l, suc_l, act_l, reset_l = [], [], [], [], [], []
        for ep in range(6):
            o = env.reset()
            reset_l.append(len(obs_l))
            obs_l.append(np.asarray(o, dtype=np.float64)); rew_l.append(0.0); done_l.append(False); suc_l.append(False); act_l.append(-1)
            # episodes alternate between a random policy and "always push right/left"
            for t in range(200):
                a = int(rng.integers(0, 3)) if ep % 3 == 0 else (2 if ep % 3 == 1 else 0)
                o, r, d, info = env.step(a)
                obs_l.append(np.asarray(o, dtype=np.float64)); rew_l.append(float(r)); done_l.append(bool(d))
                suc_l.append(bool(info["is_success"])); act_l.append(a)
                if d:
                    break
        out[f"carflag_s{seed}/obs"] = np.array(obs_l); out[f"carflag_s{seed}/rew"] = np.array(rew_l)
        out[f"carflag_s{seed}/done"] = np.array(done_l); out[f"carflag_s{seed}/success"] = np.array(suc_l)
        out[f"carflag_s{seed}/act"] = np.array(act_l); out[f"carflag_s{seed}/resets"] = np.array(reset_l)

        env = mem.Memory(num_pairs=5)
        env.seed(seed)
        obs_l, rew_l, done_l, suc_l, act_l, reset_l = [], [], [], [], [], []
        for ep in range(6):
            o = env.reset()
            reset_l.append(len(obs_l))
            obs_l.append(np.array(o, dtype=np.float64)); rew_l.append(0.0); done_l.append(False); suc_l.append(False); act_l.append(-1)
            for t in range(50):
                if ep % 2 == 0:
                    a = int(rng.integers(0, 10))
                else:   # cheating policy that reads env.state -> exercises the success branch
                    cur = env.current_card
                    cand = [i for i in range(10) if env.state[i] == env.state[cur] and i != cur]
                    a = cand[0]
                o, r, d, info = env.step(a)
                obs_l.append(np.array(o, dtype=np.float64)); rew_l.append(float(r)); done_l.append(bool(d))
                suc_l.append(bool(info.get("is_success", False))); act_l.append(a)
                if d:
                    break
        out[f"memory_s{seed}/obs"] = np.array(obs_l); out[f"memory_s{seed}/rew"] = np.array(rew_l)
        out[f"memory_s{seed}/done"] = np.array(done_l); out[f"memory_s{seed}/success"] = np.array(suc_l)
        out[f"memory_s{seed}/act"] = np.array(act_l); out[f"memory_s{seed}/resets"] = np.array(reset_l)
    np.savez_compressed(os.path.join(HERE, "G6_env_traces.npz"), **out)


def gen_G7():
    """LinearAnneal, RunningAverage and Context traces (utils/epsilon_anneal.py:28-34,
    utils/logging_utils.py:10-24, utils/context.py:36-96 incl. the int-truncation quirk)."""
    out = {"stamp": json.dumps(STAMP)}
    eps = RefAnneal(1.0, 0.1, 50)
    vals = []
    for _ in range(400):
        vals.append(eps.val)
        eps.anneal()
    out["anneal_1.0_0.1_50"] = np.array(vals)
    ra = RefRunAvg(5)
    xs = np.random.Generator(np.random.PCG64(3)).normal(size=17)
    means = []
    for x in xs:
        ra.add(float(x))
        means.append(ra.mean())
    out["runavg_in"], out["runavg_mean"] = xs, np.array(means)
    out["runavg_empty_mean"] = RefRunAvg(5).mean()
    # Context: continuous env with integer mask -5 -> int64 storage truncates floats (quirk 2)
    for tag, (mask, olen, disc) in {"cont": (-5, 3, False), "disc": (8, 2, True)}.items():
        ref_random.RNG.rng = np.random.Generator(np.random.PCG64(9))
        ctx = RefContext(4, mask, 3, olen)
        rng = np.random.Generator(np.random.PCG64(10))
        o0 = rng.uniform(-1, 1, olen) if not disc else rng.integers(0, 8, olen)
        ctx.reset(o0)
        obs_tr, act_tr, ts_tr, ins = [ctx.obs.copy()], [ctx.action.copy()], [ctx.timestep], [np.asarray(o0, dtype=np.float64)]
        for t in range(7):
            o = rng.uniform(-1, 1, olen) if not disc else rng.integers(0, 8, olen)
            a = int(rng.integers(0, 3))
            ctx.add_transition(o, a, 1.0, False)
            obs_tr.append(ctx.obs.copy()); act_tr.append(ctx.action.copy()); ts_tr.append(ctx.timestep)
            ins.append(np.concatenate([np.asarray(o, dtype=np.float64), [a]]))
        out[f"ctx_{tag}/obs"] = np.array(obs_tr); out[f"ctx_{tag}/action"] = np.array(act_tr)
        out[f"ctx_{tag}/timestep"] = np.array(ts_tr)
        out[f"ctx_{tag}/in0"] = ins[0]; out[f"ctx_{tag}/ins"] = np.array(ins[1:])
        out[f"ctx_{tag}/obs_dtype"] = str(ctx.obs.dtype)
    np.savez_compressed(os.path.join(HERE, "G7_misc.npz"), **out)


def gen_G8():
    """CSV logger rows (utils/logging_utils.py:42-109) for a scripted two-env log sequence incl. a resume (second
    logger instance appends, writes no second header), and the initial-parameter statistics of DTQN.__init__
    (utils/torch_utils.py:4-15 applied at dtqn/networks/dtqn.py:156): per state_dict entry mean / std / min / max of a
    freshly constructed reference network, for the default, a GRU / action-embedding / discrete and a sinusoidal net."""
    import argparse
    import tempfile
    from utils.logging_utils import CSVLogger as RefCSV
    out = {"stamp": json.dumps(STAMP)}
    envs = ["DiscreteCarFlag-v0", "Memory-5-v0"]
    script = []
    rng = np.random.Generator(np.random.PCG64(5))
    for i in range(3):
        row = {"losses/hours": float(rng.uniform(0, 2)), "losses/TD_Error": float(rng.uniform(0, 1)),
               "losses/Grad_Norm": float(rng.uniform(0, 5)), "losses/Max_Q_Value": float(rng.normal()),
               "losses/Mean_Q_Value": float(rng.normal()), "losses/Min_Q_Value": float(rng.normal()),
               "losses/Max_Target_Value": float(rng.normal()), "losses/Mean_Target_Value": float(rng.normal()),
               "losses/Min_Target_Value": float(rng.normal())}
        for e in envs:
            row.update({f"{e}/SuccessRate": float(rng.integers(0, 11)) / 10, f"{e}/EpisodeLength": float(rng.uniform(5, 200)),
                        f"{e}/Return": float(rng.normal())})
        script.append((row, 5000 * i))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "run")
        args = argparse.Namespace(envs=envs)
        lg = RefCSV(path, args)
        for row, step in script[:2]:
            lg.log(row, step)
        lg2 = RefCSV(path, args)                     # resume: files exist, no new header
        lg2.log(*script[2])
        out["csv_results"] = open(path + "_results.csv", newline="").read()
        out["csv_losses"] = open(path + "_losses.csv", newline="").read()
    out["csv_script"] = json.dumps([[r, s] for r, s in script])
    out["csv_envs"] = json.dumps(envs)
    # --- init statistics
    cfgs = {"default": O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50),
            "gru_a8_disc": O.NetCfg(obs_dim=10, num_actions=10, action_dim=8, inner_embed_size=128, num_heads=8, num_layers=2,
                                    history_len=50, gate="gru", discrete=True, vocab_sizes=9),
            "sin": O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=20, pos="sin")}
    stats = {}
    for name, cfg in cfgs.items():
        torch.manual_seed(3)
        net = RefDTQN(cfg.obs_dim, cfg.num_actions, cfg.embed_per_obs_dim, cfg.action_dim, cfg.inner_embed_size, cfg.num_heads,
                      cfg.num_layers, cfg.history_len, dropout=0.0, gate=cfg.gate, identity=cfg.identity, pos=cfg.pos,
                      discrete=cfg.discrete, vocab_sizes=cfg.vocab_sizes if cfg.discrete else None, bag_size=0)
        st = {}
        for k, v in net.state_dict().items():
            f = v.double()
            fin = f[torch.isfinite(f)]
            st[k] = {"shape": list(v.shape), "mean": float(fin.mean()), "std": float(fin.std(unbiased=False)) if fin.numel() > 1 else 0.0,
                     "min": float(fin.min()), "max": float(fin.max()), "requires_grad": bool(dict(net.named_parameters()).get(k, v).requires_grad)
                     if k in dict(net.named_parameters()) else None}
        stats[name] = {"cfg": cfg.to_json(), "tensors": st}
    out["init_stats"] = json.dumps(stats)
    np.savez_compressed(os.path.join(HERE, "G8_logging_init.npz"), **out)


def gen_G9():
    """Persistent-memory bag (utils/bag.py, dtqn/networks/dtqn.py:134-147,201-214, dtqn/agents/dtqn.py:116-160,166-196,
    dtqn/buffers/replay_buffer.py:171-264), from the reference itself, for a discrete and a continuous network:
      fwd_*   DTQN.forward with bag_obss / bag_actions at several sequence lengths;
      td_*    DtqnAgent.train() x2 on a synthetic buffer: the nine arrays sample_with_bag returned (Python `random` stream seeded
              right before), pre-clip gradients of update 0, statistics, parameters after update 0;
      act_*   a greedy rollout longer than the context: per step the action taken and the bag (pos, obss, actions) after
              observe() -- exercises Bag.add and the evict-by-Q-value choice.
    The continuous case uses a FLOAT padding value (-5.0): with the integer mask the reference's dtype-less np.full makes
    int64 bags / contexts that truncate the observations (utils/bag.py:42-51), a quirk pinned nowhere else."""
    out = {"stamp": json.dumps(STAMP)}
    cases = [("disc", O.NetCfg(obs_dim=2, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=10, discrete=True,
                               vocab_sizes=7, action_dim=8, bag_size=4), 6, 40),
             ("cont", O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=12, bag_size=5),
              -5.0, 40)]
    out["names"] = json.dumps([c[0] for c in cases])
    for name, cfg, mask, T in cases:
        seed, B, n_eps = 50 + len(name), 6, 14
        out[f"{name}_cfg"] = json.dumps(cfg.to_json())
        out[f"{name}_meta"] = json.dumps({"seed": seed, "B": B, "T": T, "n_eps": n_eps, "mask": mask})
        pol = O.init_params(cfg, seed=seed, perturb=True)
        tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
        out[f"{name}_pol_checksum"] = checksum(pol)
        ot = torch.long if cfg.discrete else torch.float32
        rng = np.random.Generator(np.random.PCG64(seed + 500))
        draw = lambda *shape: (rng.integers(0, cfg.vocab_sizes - 1, size=shape).astype(np.int64) if cfg.discrete
                               else rng.uniform(-1, 1, size=shape).astype(np.float32))
        # ---- forward
        net = make_ref_net(cfg, pol)
        for n in (1, cfg.history_len // 2, cfg.history_len):
            obs, bag_obs = draw(3, n, cfg.obs_dim), draw(3, cfg.bag_size, cfg.obs_dim)
            act = rng.integers(0, cfg.num_actions, size=(3, n, 1))
            bag_act = rng.integers(0, cfg.num_actions, size=(3, cfg.bag_size, 1))
            with torch.no_grad():
                q = net(torch.as_tensor(obs, dtype=ot), torch.as_tensor(act), torch.as_tensor(bag_obs, dtype=ot), torch.as_tensor(bag_act)).numpy()
            out.update({f"{name}_fwd{n}_obs": obs, f"{name}_fwd{n}_act": act, f"{name}_fwd{n}_bag_obs": bag_obs,
                        f"{name}_fwd{n}_bag_act": bag_act, f"{name}_fwd{n}_q": q})
        # ---- TD updates
        ref_random.RNG.rng = np.random.Generator(np.random.PCG64(seed))
        agent = make_ref_agent(cfg, pol, tgt, B, T, n_eps + 2, mask, tuf=10_000)
        episodes = synth_episodes(rng, n_eps, T, cfg, min_len=cfg.history_len + 3)
        fill_agent(agent, episodes)
        agent.eval_off()
        random.seed(seed + 7)
        rec = run_ref_updates(agent, 2)
        names9 = ["obss", "actions", "rewards", "next_obss", "next_actions", "dones", "ep_lens", "bag_obss", "bag_actions"]
        for i, b in enumerate(rec["batches"]):
            assert len(b) == 9
            out.update({f"{name}_td_batch{i}_{k}": np.asarray(a) for k, a in zip(names9, b)})
        for j, (obs, act, rew, done) in enumerate(episodes):
            out.update({f"{name}_ep{j}_obs": obs, f"{name}_ep{j}_act": act, f"{name}_ep{j}_rew": rew})
        keys = O.trainable_keys(cfg)
        pnames = [n for n, p in agent.policy_network.named_parameters()]
        gl = {n: g for n, g in zip(pnames, rec["grads"][0]) if g is not None}
        assert sorted(gl) == sorted(keys)
        out[f"{name}_td_grad0_flat"] = np.concatenate([gl[k].numpy().ravel() for k in keys])
        out[f"{name}_td_post0_flat"] = rec["post"][0]
        out[f"{name}_td_stats"] = json.dumps(rec["stats"])
        out[f"{name}_td_grad_norms"] = np.array(rec["norms"], dtype=np.float64)
        # ---- greedy rollout with bag evictions
        ref_random.RNG.rng = np.random.Generator(np.random.PCG64(seed + 3))
        agent = make_ref_agent(cfg, pol, tgt, B, T, n_eps + 2, mask)
        agent.eval_off()
        steps = cfg.history_len + 3 * cfg.bag_size + 2
        traj = draw(steps + 1, cfg.obs_dim)
        agent.context_reset(traj[0])
        acts, poss, bobs, bacts = [], [], [], []
        for t in range(steps):
            a = int(agent.get_action(epsilon=0.0))
            agent.observe(traj[t + 1], a, 0.0, False)
            acts.append(a)
            poss.append(agent.bag.pos)
            bobs.append(np.array(agent.bag.obss, dtype=np.float64))
            bacts.append(np.array(agent.bag.actions, dtype=np.int64))
        out.update({f"{name}_act_traj": traj, f"{name}_act_actions": np.array(acts), f"{name}_act_bag_pos": np.array(poss),
                    f"{name}_act_bag_obss": np.stack(bobs), f"{name}_act_bag_actions": np.stack(bacts)})
    np.savez_compressed(os.path.join(HERE, "G9_bag.npz"), **out)


class RefDropoutHash:
    """Stands in for torch.nn.functional.dropout while THE REFERENCE runs (gen_G10).  torch's Philox stream cannot be matched
    by another implementation, but everything else about the reference's dropout can be pinned: WHICH tensors are dropped (the
    call sites), in which forward passes (train / eval mode, i.e. whether `training` arrives True), and how survivors are scaled.
    Each call the reference makes with training=True is answered with the keep mask of oracle.dtqn_oracle.drop_keep for
    (seed, step, pass, sequence, site, layer, element); the site is identified by the ORDER and SHAPE of the reference's own
    calls inside one DTQN.forward (dtqn/networks/dtqn.py:196 -> [B, n, D]; per layer transformer.py:34 attention weights
    [B*H, n, n] then transformer.py:41 [B, n, D]; dtqn.py:136-141 bag attention weights [B*H, n, bag]).  A call the restatement
    does not expect (wrong shape, wrong order, an extra site) raises, so the fixture cannot be written from a different
    site structure than the one oracle / kernels implement."""

    def __init__(self, cfg: O.NetCfg, seed: int):
        self.cfg, self.seed, self.step, self.which, self.calls, self.log = cfg, seed, 0, None, 0, []

    def begin_pass(self, which):
        self.which, self.calls = which, 0

    def __call__(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            self.log.append(("eval", self.which, tuple(input.shape)))
            return input
        cfg = self.cfg
        assert self.which is not None, "train-mode dropout outside a tracked policy forward"
        spec = O.DropSpec(float(p), self.seed, self.step, self.which)
        k = self.calls
        self.calls += 1
        H, D, NL = cfg.num_heads, cfg.inner_embed_size, cfg.num_layers
        if k == 0:
            assert input.dim() == 3 and input.shape[-1] == D, input.shape
            self.log.append(("emb", self.which, tuple(input.shape)))
            return O.drop_rows(spec, input, O.DROP_EMB, 0)
        j, layer = (k - 1) % 2, (k - 1) // 2
        if layer < NL and j == 0:
            BH, n, m = input.shape
            assert BH % H == 0 and n == m, input.shape
            self.log.append(("attn", self.which, layer, tuple(input.shape)))
            return O.drop_attn(spec, input.reshape(BH // H, H, n, m), layer).reshape(BH, n, m)
        if layer < NL:
            assert input.dim() == 3 and input.shape[-1] == D, input.shape
            self.log.append(("ffn", self.which, layer, tuple(input.shape)))
            return O.drop_rows(spec, input, O.DROP_FFN, layer)
        assert cfg.bag_size > 0 and k == 1 + 2 * NL, ("unexpected dropout call", k, tuple(input.shape))
        BH, n, m = input.shape
        assert m == cfg.bag_size and BH % H == 0, input.shape
        self.log.append(("bag", self.which, tuple(input.shape)))
        Bn = BH // H
        idx = ((np.arange(H)[:, None, None] << 16) | (np.arange(n)[None, :, None] << 8) | np.arange(m)[None, None, :]).astype(np.uint64)
        keep = np.stack([O.drop_keep(spec, b, O.DROP_BAG, 0, idx) for b in range(Bn)])
        return input * torch.from_numpy(keep.astype(np.float32) * np.float32(spec.scale)).reshape(BH, n, m)


def gen_G10():
    """Dropout pinned to the reference.  THE REFERENCE's DtqnAgent.train() runs with --dropout p on
    res / GRU / identity / bag networks while torch.nn.functional.dropout is replaced by RefDropoutHash: the reference decides
    where and when dropout applies, the hash only supplies reproducible keep masks.  Stored per case and update: the sampled
    batch, the Q-values of the three forwards AS train() COMPUTED THEM (policy(o) and policy(o') in train mode with independent
    masks, target(o') in eval mode), pre-clip gradients, statistics, parameters after the step, and the log of the reference's
    dropout calls (site order, shapes, and which calls arrived with training=False)."""
    import torch.nn.functional as F
    cases = [
        ("res", O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50, dropout=0.1), 8, 120, -5),
        ("gru", O.NetCfg(obs_dim=3, num_actions=4, inner_embed_size=32, num_heads=4, num_layers=2, history_len=20, gate="gru", action_dim=4,
                         dropout=0.2), 4, 40, -5),
        ("ident", O.NetCfg(obs_dim=6, num_actions=5, inner_embed_size=32, num_heads=4, num_layers=2, history_len=24, identity=True, pos="sin",
                           discrete=True, vocab_sizes=9, dropout=0.15), 4, 40, 8),
        ("bag", O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=1, history_len=12, bag_size=5, dropout=0.1),
         6, 40, -5.0),
    ]
    out = {"stamp": json.dumps(STAMP), "names": json.dumps([c[0] for c in cases]), "n_updates": 2}
    orig_dropout = F.dropout
    for name, cfg, B, T, mask in cases:
        seed, drop_seed, n_eps = 300 + len(name), 4242, 14
        random.seed(seed)
        ref_random.RNG.rng = np.random.Generator(np.random.PCG64(seed))
        rng = np.random.Generator(np.random.PCG64(seed + 1000))
        pol = O.init_params(cfg, seed=seed, perturb=True)
        tgt = O.init_params(cfg, seed=seed + 1, perturb=True)
        agent = make_ref_agent(cfg, pol, tgt, B, T, n_eps + 2, mask)
        fill_agent(agent, synth_episodes(rng, n_eps, T, cfg, min_len=cfg.history_len + 3 if cfg.bag_size else 3))
        hasher = RefDropoutHash(cfg, drop_seed)
        qs = []
        n_pol = [0]

        def pol_pre(mod, args):
            hasher.begin_pass(n_pol[0] % 2)          # train(): policy(o) is pass 0, policy(o') pass 1 (dtqn.py:215,226)
            n_pol[0] += 1

        def tgt_pre(mod, args):
            hasher.begin_pass(None)                  # eval mode: a training=True call here would trip the assert

        keep_q = lambda mod, args, outp: qs.append(outp.detach().numpy().copy())
        hooks = [agent.policy_network.register_forward_pre_hook(pol_pre), agent.target_network.register_forward_pre_hook(tgt_pre),
                 agent.policy_network.register_forward_hook(keep_q), agent.target_network.register_forward_hook(keep_q)]
        F.dropout = hasher
        try:
            agent.eval_off()
            random.seed(seed + 7)
            recs, draws = [], []
            orig_choice, orig_randint = random.choice, random.randint
            for it in range(2):
                hasher.step = it
                log = {"choice": [], "randint": []}
                # the sampler's draws (replay_buffer.py:146-157 / :186-205): B episode choices, then B window starts
                random.choice = lambda seq, _l=log: (_l["choice"].append(orig_choice(seq)) or _l["choice"][-1])
                random.randint = lambda a, b, _l=log: (_l["randint"].append(orig_randint(a, b)) or _l["randint"][-1])
                try:
                    recs.append(run_ref_updates(agent, 1))
                finally:
                    random.choice, random.randint = orig_choice, orig_randint
                draws.append(log)
        finally:
            F.dropout = orig_dropout
            for h in hooks:
                h.remove()
        assert len(qs) == 6 and n_pol[0] == 4
        assert agent.policy_network.training and not agent.target_network.training
        out[f"{name}_cfg"] = json.dumps(cfg.to_json())
        out[f"{name}_meta"] = json.dumps({"seed": seed, "drop_seed": drop_seed, "B": B, "T": T, "n_eps": n_eps, "mask": mask})
        out[f"{name}_pol_checksum"] = checksum(pol)
        out[f"{name}_drop_calls"] = json.dumps(hasher.log)
        keys = O.trainable_keys(cfg)
        pnames = [n for n, p in agent.policy_network.named_parameters()]
        names9 = ["obss", "actions", "rewards", "next_obss", "next_actions", "dones", "ep_lens", "bag_obss", "bag_actions"]
        for it, rec in enumerate(recs):
            out.update({f"{name}_u{it}_{k}": np.asarray(a) for k, a in zip(names9, rec["batches"][0])})
            assert len(draws[it]["choice"]) == B and len(draws[it]["randint"]) == B
            out[f"{name}_u{it}_ep_idx"] = np.array(draws[it]["choice"], dtype=np.int32)
            out[f"{name}_u{it}_start"] = np.array(draws[it]["randint"], dtype=np.int32)
            out[f"{name}_u{it}_q_all"], out[f"{name}_u{it}_q_next_pol"], out[f"{name}_u{it}_q_next_tgt"] = qs[3 * it:3 * it + 3]
            gl = {n: g for n, g in zip(pnames, rec["grads"][0]) if g is not None}
            assert sorted(gl) == sorted(keys)
            out[f"{name}_u{it}_grad_flat"] = np.concatenate([gl[k].numpy().ravel() for k in keys])
            if it == 0:      # update 1 starts from these (pre of update 0 = the seeded parameters)
                out[f"{name}_u{it}_post_flat"] = rec["post"][0]
            out[f"{name}_u{it}_stats"] = json.dumps(rec["stats"][0])
            out[f"{name}_u{it}_grad_norm"] = rec["norms"][0]
        # the episodes, so that a device replay can be filled with the same content (bag sampling reads rows before the window)
        out[f"{name}_replay_obss"] = np.asarray(agent.replay_buffer.obss)
        out[f"{name}_replay_actions"] = np.asarray(agent.replay_buffer.actions)
        out[f"{name}_replay_rewards"] = np.asarray(agent.replay_buffer.rewards)
        out[f"{name}_replay_dones"] = np.asarray(agent.replay_buffer.dones)
        out[f"{name}_replay_lens"] = np.asarray(agent.replay_buffer.episode_lengths)
        print(name, "dropout calls:", len(hasher.log), "eval-mode calls:", sum(1 for c in hasher.log if c[0] == "eval"))
    np.savez_compressed(os.path.join(HERE, "G10_dropout.npz"), **out)


def gen_G11():
    """Image observations (dtqn/networks/representations.py:77-130 reached from dtqn/networks/dtqn.py:71-77): the REFERENCE's DTQN built
    with obs_dim = (C, H, W), on uint8 pixel windows cast to float32 unscaled as DtqnAgent.train() does (dtqn/agents/dtqn.py:199-202).
      td_*   two shapes: the three forwards of a TD update, the loss of dtqn.py:219-243 evaluated with the reference modules, and its
             gradient w.r.t. every parameter (autograd through the reference network);
      fwd144 one forward at the MiniHack pixel-crop size 3 x 144 x 144 (obs_crop 9 x 16-pixel tiles, envs/mini_hack.py:18-76)."""
    out = {"stamp": json.dumps(STAMP)}
    cases = [("a", O.NetCfg(obs_dim=3 * 16 * 16, num_actions=4, inner_embed_size=64, num_heads=4, num_layers=1, history_len=6, image=(3, 16, 16)), 3),
             ("b", O.NetCfg(obs_dim=1 * 21 * 13, num_actions=3, inner_embed_size=128, num_heads=8, num_layers=2, history_len=5, image=(1, 21, 13),
                            pos="sin"), 2)]
    out["names"] = json.dumps([c[0] for c in cases])
    for name, cfg, B in cases:
        seed, L, A = 400 + len(out), cfg.history_len, cfg.num_actions
        rng = np.random.Generator(np.random.PCG64(seed))
        pol, tgt = O.init_params(cfg, seed=seed, perturb=True), O.init_params(cfg, seed=seed + 1, perturb=True)
        npol, ntgt = make_ref_net(cfg, pol), make_ref_net(cfg, tgt)
        npol.train(); ntgt.eval()
        rows = rng.integers(0, 256, size=(B, L + 1, *cfg.image)).astype(np.uint8)
        rows[0, L - 1:] = 0                                          # a padded tail (obs_mask 0 for images, env_processing.py:106-108)
        acts = rng.integers(0, A, size=(B, L + 1, 1))
        rew = rng.choice(np.array([0, 0, 1, -1], dtype=np.float32), size=(B, L, 1))
        done = (rng.random((B, L, 1)) < 0.2)
        o, o2 = torch.as_tensor(rows[:, :L], dtype=torch.float32), torch.as_tensor(rows[:, 1:], dtype=torch.float32)
        a, a2 = torch.as_tensor(acts[:, :L]), torch.as_tensor(acts[:, 1:])
        q_all = npol(o, a)                                           # dtqn.py:215
        q_sel = q_all.gather(2, a).squeeze()
        with torch.no_grad():
            q_np = npol(o2, a2)                                      # :226
            amax = torch.argmax(q_np, dim=2).unsqueeze(-1)
            q_nt = ntgt(o2, a2)                                      # :230
            nq = q_nt.gather(2, amax).squeeze()
            targets = torch.as_tensor(rew).squeeze() + (1 - torch.as_tensor(done, dtype=torch.long).squeeze()) * (nq * 0.99)
        loss = torch.nn.functional.mse_loss(q_sel, targets)          # :243 (history == context)
        loss.backward()
        keys = O.trainable_keys(cfg)
        named = dict(npol.named_parameters())
        assert sorted(k for k, p in named.items() if p.requires_grad) == sorted(keys)
        out.update({f"{name}_cfg": json.dumps(cfg.to_json()), f"{name}_seed": seed, f"{name}_B": B, f"{name}_pol_checksum": checksum(pol),
                    f"{name}_rows": rows, f"{name}_actions": acts, f"{name}_rewards": rew, f"{name}_dones": done,
                    f"{name}_q_all": q_all.detach().numpy(), f"{name}_q_next_pol": q_np.numpy(), f"{name}_q_next_tgt": q_nt.numpy(),
                    f"{name}_loss": float(loss), f"{name}_grad_flat": np.concatenate([named[k].grad.numpy().ravel() for k in keys])})
    cfg = O.NetCfg(obs_dim=3 * 144 * 144, num_actions=8, inner_embed_size=64, num_heads=8, num_layers=2, history_len=4, image=(3, 144, 144))
    pol = O.init_params(cfg, seed=440, perturb=True)
    rng = np.random.Generator(np.random.PCG64(441))
    obs = rng.integers(0, 256, size=(1, 3, 3, 144, 144)).astype(np.uint8)
    with torch.no_grad():
        q = make_ref_net(cfg, pol)(torch.as_tensor(obs, dtype=torch.float32), torch.zeros(1, 3, 1, dtype=torch.long)).numpy()
    out.update({"fwd144_cfg": json.dumps(cfg.to_json()), "fwd144_seed": 440, "fwd144_obs": obs, "fwd144_q": q, "fwd144_pol_checksum": checksum(pol)})
    np.savez_compressed(os.path.join(HERE, "G11_image.npz"), **out)


# --------------------------------------------------------------------------- #
# G12: the coupled loop (run.py) -- THE REFERENCE's own run.py functions, imported unmodified
# --------------------------------------------------------------------------- #
def install_run_stubs():
    """What /root/reference/run.py, utils/env_processing.py and envs/__init__.py import on top of install_stubs():
    `gym.wrappers.time_limit.TimeLimit` (third-party gym 0.18.0, requirements.txt; its step-count semantics are restated
    here: count steps since reset, at the cap set done and info["TimeLimit.truncated"] = not done) and
    `gym.envs.registration.register` (a no-op: the envs are built directly below, as envs/__init__.py:31-48 registers them)."""
    gym = sys.modules["gym"]
    gym.__path__ = []

    class TimeLimit(gym.Wrapper):
        def __init__(self, env, max_episode_steps=None):
            super().__init__(env)
            self._max_episode_steps, self._elapsed_steps = max_episode_steps, None
            self.observation_space, self.action_space = env.observation_space, env.action_space

        def seed(self, seed=None):
            return self.env.seed(seed)

        def reset(self, **kw):
            self._elapsed_steps = 0
            return self.env.reset(**kw)

        def step(self, action):
            obs, reward, done, info = self.env.step(action)
            self._elapsed_steps += 1
            if self._elapsed_steps >= self._max_episode_steps:
                info["TimeLimit.truncated"] = not done
                done = True
            return obs, reward, done, info

    wr = types.ModuleType("gym.wrappers")
    wr.__path__ = []
    tl = types.ModuleType("gym.wrappers.time_limit")
    tl.TimeLimit = TimeLimit
    wr.time_limit = tl
    ge = types.ModuleType("gym.envs")
    ge.__path__ = []
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **kw: None
    sys.modules.update({"gym.wrappers": wr, "gym.wrappers.time_limit": tl, "gym.envs": ge, "gym.envs.registration": reg})
    return TimeLimit


class _DrawRecorder:
    """Stands in for the `random` module inside dtqn/buffers/replay_buffer.py: same stream, draws logged."""

    def __init__(self):
        self.choices, self.randints = [], []

    def choice(self, seq):
        v = random.choice(seq)
        self.choices.append(int(v))
        return v

    def randint(self, a, b):
        v = random.randint(a, b)
        self.randints.append(int(v))
        return v

    def __getattr__(self, name):
        return getattr(random, name)


def loop_case(ref_run, make_env, *, seed, D, H, NL, L, B, buf_size, prepop, steps, tuf, eval_frequency, eval_episodes, lr=3e-4):
    """run_experiment (run.py:408-523) for one DiscreteCarFlag run, with run.py's OWN set_global_seed / get_agent / prepopulate /
    train (hence step and evaluate) and the weights replaced by oracle.init_params so that the other side can regenerate them.
    Everything the loop does is logged from the outside: every get_action (epsilon, action, Q of the last row when greedy),
    every observe, the sampler's draws and the statistics of every update, the logger rows."""
    import dtqn.buffers.replay_buffer as rbmod
    envs, eval_envs = [make_env()], [make_env()]
    ref_run.set_global_seed(seed, *(envs + eval_envs))                      # run.py:418
    eps = ref_run.epsilon_anneal.LinearAnneal(1.0, 0.1, steps // 10)        # run.py:420
    agent = ref_run.get_agent("DTQN", envs, 8, 0, D, buf_size, torch.device("cpu"), lr, B, L, -1, L, tuf, 0.99, H, NL, 0.0, False,
                              "res", "learned", 0)                         # run.py:422-445 (positional, like the reference)
    agent.replay_buffer.episode_lengths = agent.replay_buffer.episode_lengths.astype(np.int64)   # quirk 1
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=D, num_heads=H, num_layers=NL, history_len=L)
    pol = O.init_params(cfg, seed=seed + 100, perturb=True)
    assert list(agent.policy_network.state_dict().keys()) == O.state_dict_keys(cfg)
    agent.policy_network.load_state_dict({k: v.clone() for k, v in pol.items()})
    agent.target_update()                                                   # DqnAgent.__init__'s hard copy (dqn.py:49), redone
    ref_run.prepopulate(agent, prepop, envs)                                # run.py:499 (there with 50 000 steps)
    rb = agent.replay_buffer
    n_used = min(rb.pos[0] + 1, rb.max_size)
    out = {"cfg": json.dumps(cfg.to_json()), "seed": seed, "B": B, "buf_size": buf_size, "prepop": prepop, "steps": steps, "tuf": tuf,
           "lr": lr, "eval_frequency": eval_frequency, "eval_episodes": eval_episodes, "pol_seed": seed + 100,
           "pol_checksum": checksum(pol), "stamp": json.dumps(STAMP),
           "prepop/pos": np.array(rb.pos), "prepop/obss": rb.obss[:n_used].copy(), "prepop/actions": rb.actions[:n_used, :, 0].copy(),
           "prepop/rewards": rb.rewards[:n_used, :, 0].copy(), "prepop/dones": rb.dones[:n_used, :, 0].copy(),
           "prepop/eplens": rb.episode_lengths[:n_used].copy(), "prepop/rng_probe": ref_random.RNG.rng.bit_generator.state["state"]["state"] % (1 << 53)}
    # ---- instrumentation (outside the reference's code) ----
    ev = {"act_mode": [], "act_eps": [], "act_action": [], "act_greedy": [], "act_q": [], "obs_mode": [], "obs_obs": [], "obs_action": [],
          "obs_reward": [], "obs_done": [], "upd_ep": [], "upd_start": [], "upd_stats": [], "upd_after_act": [], "flush_after_obs": []}
    state = {"in_action": False, "q": None}
    agent.policy_network.register_forward_hook(lambda m, i, o: state.__setitem__("q", o.detach()[0, -1].numpy().copy()) if state["in_action"] else None)
    orig_get, orig_obs, orig_train, orig_flush = agent.get_action, agent.observe, agent.train, rb.flush
    is_eval = lambda: int(agent.train_mode.name == "EVAL")

    def get_action(epsilon=0.0):
        state["in_action"], state["q"] = True, None
        a = orig_get(epsilon=epsilon)
        state["in_action"] = False
        ev["act_mode"].append(is_eval()); ev["act_eps"].append(float(epsilon)); ev["act_action"].append(int(a))
        ev["act_greedy"].append(state["q"] is not None)
        ev["act_q"].append(state["q"] if state["q"] is not None else np.full(3, np.nan, dtype=np.float32))
        return a

    def observe(obs, action, reward, done):
        ev["obs_mode"].append(is_eval()); ev["obs_obs"].append(np.asarray(obs, dtype=np.float64).copy()); ev["obs_action"].append(int(action))
        ev["obs_reward"].append(float(reward)); ev["obs_done"].append(bool(done))
        return orig_obs(obs, action, reward, done)

    rec = _DrawRecorder()
    rbmod.random = rec

    def train():
        n0 = agent.num_train_steps
        rec.choices.clear(); rec.randints.clear()
        orig_train()
        if agent.num_train_steps != n0:
            ev["upd_ep"].append(np.array(rec.choices)); ev["upd_start"].append(np.array(rec.randints))
            ev["upd_after_act"].append(len(ev["act_action"]))
            ev["upd_stats"].append([agent.td_errors.q[-1], agent.grad_norms.q[-1], agent.qvalue_max.q[-1], agent.qvalue_mean.q[-1],
                                    agent.qvalue_min.q[-1], agent.target_max.q[-1], agent.target_mean.q[-1], agent.target_min.q[-1]])

    def flush():
        ev["flush_after_obs"].append(len(ev["obs_action"]))
        return orig_flush()

    agent.get_action, agent.observe, agent.train, rb.flush = get_action, observe, train, flush
    rows = []

    class Logger:
        def log(self, results, step):
            rows.append((int(step), {k: float(v) for k, v in results.items() if k != "losses/hours"}))

    RA = ref_run.RunningAverage
    try:
        ref_run.train(agent, envs, eval_envs, ["DiscreteCarFlag-v0"], steps, eps, eval_frequency, eval_episodes, "/nonexistent/policy", False,
                      Logger(), RA(10), RA(10), RA(10), None, False)           # run.py:503-520
    finally:
        rbmod.random = random
    out.update({f"ev/{k}": np.array(v) for k, v in ev.items()})
    out["log_steps"] = np.array([s for s, _ in rows])
    out["log_rows"] = json.dumps([r for _, r in rows])
    keys = O.trainable_keys(cfg)
    sd = agent.policy_network.state_dict()
    out["final_flat"] = np.concatenate([sd[k].numpy().ravel() for k in keys])
    out["final/pos"] = np.array(rb.pos)
    out["final/eps"] = float(eps.val)
    out["final/num_train_steps"] = int(agent.num_train_steps)
    out["final/rng_probe"] = ref_random.RNG.rng.bit_generator.state["state"]["state"] % (1 << 53)
    return out


def gen_G12():
    """The coupled actor / learner loop: run.py:287-298 (step -> flush / reset -> train -> anneal), :356-377 (step), :380-405
    (prepopulate), :187-243 (evaluate, at timestep % eval_frequency == 0), driven by run.py's own code on the reference's CarFlag
    (envs/car_flag.py, discrete, 200-step cap as envs/__init__.py:44-48 registers it).  Pins WHICH RNG stream is consumed WHEN
    (env resets inside get_agent's space probes, context padding draws, epsilon draws, sampler draws), when the first update happens,
    what the buffer holds, and the evaluation interleave."""
    TimeLimit = install_run_stubs()
    import run as ref_run
    assert os.path.realpath(ref_run.__file__) == os.path.join(REF, "run.py")
    car = load_by_path("ref_car_flag", os.path.join(REF, "envs/car_flag.py"))
    make_env = lambda: TimeLimit(car.CarFlag(discrete=True), max_episode_steps=200)
    out = {}
    cases = {"small": dict(seed=1, D=16, H=2, NL=2, L=8, B=4, buf_size=12 * 200, prepop=1200, steps=150, tuf=25, eval_frequency=60, eval_episodes=2),
             "cfg1": dict(seed=1, D=64, H=8, NL=2, L=50, B=32, buf_size=500_000, prepop=10_000, steps=200, tuf=50, eval_frequency=100, eval_episodes=2)}
    for name, kw in cases.items():
        res = loop_case(ref_run, make_env, **kw)
        out.update({f"{name}/{k}": v for k, v in res.items()})
        print(name, "updates", res["final/num_train_steps"], "actions", len(res["ev/act_action"]), "greedy", int(np.sum(res["ev/act_greedy"])))
        # The loop is free-running: every update starts from the previous one's parameters, and Adam turns noise-floor gradient
        # differences into +-lr steps, so two correct fp32 implementations drift apart.  How fast is a property of the trajectory, and
        # the reference measures it itself: the SAME run with one torch thread (only the summation order inside the CPU kernels
        # changes).  The trace of that twin is stored as `<name>_t1/*`; tests bound their distance to the reference by the
        # reference's distance to its own twin (tests/loop_harness.py).
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            twin = loop_case(ref_run, make_env, **kw)
        finally:
            torch.set_num_threads(nthreads)
        for k in ("ev/act_action", "ev/act_greedy", "ev/act_q", "ev/upd_stats", "ev/upd_after_act"):
            out[f"{name}_t1/{k}"] = twin[k]
        a, b = np.array(res["ev/act_action"]), np.array(twin["ev/act_action"])
        n = min(len(a), len(b))
        d = np.nonzero(a[:n] != b[:n])[0]
        print(name, "twin (1 thread): first differing action event", d[:1], "of", n, "threads", nthreads)
    # the non-finite branch of dtqn/agents/dtqn.py:257-261 (clip_grad_norm_(error_if_nonfinite=True)): the reference's own exception
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=16, num_heads=2, num_layers=2, history_len=8)
    random.seed(5)
    ref_random.RNG.rng = np.random.Generator(np.random.PCG64(5))
    pol = O.init_params(cfg, seed=5, perturb=True)
    agent = make_ref_agent(cfg, pol, pol, 4, 20, 10, -5)
    fill_agent(agent, synth_episodes(np.random.Generator(np.random.PCG64(6)), 8, 20, cfg))
    agent.replay_buffer.rewards[:] = np.inf
    pre = np.concatenate([p.detach().numpy().ravel() for p in agent.policy_network.parameters()])
    try:
        agent.train()
        raise AssertionError("the reference did not raise")
    except RuntimeError as e:
        out["nonfinite/message"] = str(e)
        out["nonfinite/type"] = type(e).__name__
    post = np.concatenate([p.detach().numpy().ravel() for p in agent.policy_network.parameters()])
    out["nonfinite/params_untouched"] = bool(np.array_equal(pre, post))
    out["nonfinite/num_train_steps"] = int(agent.num_train_steps)
    out["nonfinite/td_errors_len"] = len(agent.td_errors.q)          # the loss was logged before the clip raised (dtqn.py:253)
    print("nonfinite:", out["nonfinite/message"])
    np.savez_compressed(os.path.join(HERE, "G12_loop.npz"), **out)


def time_reference():
    """the reference's OWN DtqnAgent.train() on this container's CPU cores, BASELINE
    configs 1-5 (synthetic replay of SURVEY.md section 8d; configs 3-5 at their per-GPU batch, a handful of updates
    each -- one update of cfg 3 is seconds of CPU), threads = 8 and 1, median + p10 / p90.  -> ref_cpu_timing.json"""
    res = {"stamp": STAMP, "nproc": os.cpu_count(), "cpu": "", "runs": []}
    try:
        res["cpu"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    cases = [
        ("config1", O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50), 32, 200, -5, (10, 100)),
        ("config2", O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50), 256, 200, -5, (3, 20)),
        ("config3", O.NetCfg(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=8), 512, 50, 7, (1, 5)),
        ("config4", O.NetCfg(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2, history_len=128, discrete=True, vocab_sizes=12), 128, 250, 11, (1, 4)),
        ("config5", O.NetCfg(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, num_layers=2, history_len=256, discrete=True, vocab_sizes=22), 32, 256, 21, (1, 4)),
    ]
    only = os.environ.get("REF_TIME_ONLY")
    for name, cfg, B, T, mask, (n_warm, n) in cases:
        if only and name not in only.split(","):
            continue
        for threads in (8, 1):
            if threads == 1 and name in ("config3", "config4", "config5"):
                n_warm, n = 1, 2
            torch.set_num_threads(threads)
            random.seed(1)
            ref_random.RNG.rng = np.random.Generator(np.random.PCG64(1))
            rng = np.random.Generator(np.random.PCG64(1))
            pol = O.init_params(cfg, seed=1)
            n_eps = max(B + 8, 290) if T <= 64 else 290
            agent = make_ref_agent(cfg, pol, pol, B, T, n_eps + 10, mask)
            fill_agent(agent, synth_episodes(rng, n_eps, T, cfg, min_len=5))
            for _ in range(n_warm):
                agent.train()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                agent.train()
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts) * 1e3
            res["runs"].append({"config": name, "workload": f"L={cfg.history_len} D={cfg.inner_embed_size} B={B} T={T}", "threads": threads,
                                "updates": n, "ms_median": float(np.median(ts)), "ms_p10": float(np.percentile(ts, 10)),
                                "ms_p90": float(np.percentile(ts, 90)), "td_updates_per_s": float(1e3 / np.median(ts))})
            print(res["runs"][-1], flush=True)
            with open(os.path.join(HERE, "ref_cpu_timing.json"), "w") as f:
                json.dump(res, f, indent=1)
    torch.set_num_threads(8)


if __name__ == "__main__":
    which = sys.argv[1:] or ["G1", "G2", "G3", "G4", "G5", "G6", "G7", "G8", "G9", "G10", "G11", "G12", "time"]
    torch.manual_seed(0)
    for w in which:
        t0 = time.time()
        {"G1": gen_G1, "G2": gen_G2, "G3": gen_G3, "G4": gen_G4, "G5": gen_G5, "G6": gen_G6, "G7": gen_G7,
         "G8": gen_G8, "G9": gen_G9, "G10": gen_G10, "G11": gen_G11, "G12": gen_G12, "time": time_reference}[w]()
        print(f"{w}: done in {time.time() - t0:.1f}s")
