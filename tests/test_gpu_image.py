"""-m gpu: the image observation embedding on the MI355X against the reference's outputs (tests/golden/G11_image.npz, see
test_image_golden.py), the agent surface on pixel observations, and full-size properties at the MiniHack crop size."""
import numpy as np
import pytest
import torch

from test_image_golden import NAMES, check_engine_vs_g11, check_module_forward_vs_g11

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


@pytest.mark.parametrize("name", NAMES)
def test_engine_on_the_gpu_matches_the_reference_with_image_observations(lib, name):
    check_engine_vs_g11(lib, name, device="cuda", test_lib=False)


def test_module_forward_on_the_gpu_incl_the_minihack_crop(lib):
    check_module_forward_vs_g11(None, device="cuda")


class _PixelEnv:
    """Synthetic image POMDP behind the env surface run.py uses (MiniHack itself is not installable here): a bright square walks
    over a dark 3 x 32 x 32 frame; action 0 / 1 moves it left / right, reaching the right edge pays +1."""

    def __init__(self, seed=0):
        from dtqn_amd.envs import spaces
        self.observation_space = spaces.Box(low=0, high=255, shape=(3, 32, 32), dtype=np.uint8)
        self.action_space = spaces.Discrete(2)
        self._max_episode_steps = 24
        self.rng = np.random.default_rng(seed)

    def _obs(self):
        o = np.zeros((3, 32, 32), dtype=np.uint8)
        o[:, 12:20, self.x:self.x + 4] = 200 + self.rng.integers(0, 50)
        return o

    def reset(self):
        self.x, self.t = int(self.rng.integers(4, 20)), 0
        return self._obs()

    def step(self, action):
        self.x = int(np.clip(self.x + (2 if action == 1 else -2), 0, 28))
        self.t += 1
        done = self.x >= 28 or self.t >= self._max_episode_steps
        return self._obs(), (1.0 if self.x >= 28 else 0.0), done, {"TimeLimit.truncated": self.t >= self._max_episode_steps and self.x < 28}

    def seed(self, seed=None):
        self.rng = np.random.default_rng(seed)
        return [seed]


def test_agent_trains_on_pixel_observations(lib):
    """get_agent / context_reset / get_action / observe / train on an image env: uint8 replay on the device, convolutional embedding
    in front of the row-block update, parameters of the convolutions move, checkpoint round trip of the conv keys."""
    import run as runpy
    from dtqn_amd.utils.agent_utils import get_agent
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.random import set_global_seed
    env = _PixelEnv(1)
    set_global_seed(3, env)
    agent = get_agent("DTQN", [env], 8, 0, 64, 4000, torch.device("cuda"), 3e-4, 4, 8, -1, 8, 1000, 0.99, 4, 1, 0.0, False, "res", "learned", 0,
                      sampler="device", sample_seed=3)
    assert agent.image == (3, 32, 32) and agent.replay_buffer.dev.obs.dtype == torch.uint8
    runpy.prepopulate(agent, 400, [env])
    sd0 = {k: v.clone() for k, v in agent.policy_network.state_dict().items()}
    agent.context_reset(env.reset())
    for _ in range(12):
        if runpy.step(agent, env, Constant(0.3)):
            agent.replay_buffer.flush(); agent.context_reset(env.reset())
        agent.train()
    assert agent.num_train_steps == 12 and np.isfinite(agent.td_errors.mean()) and np.isfinite(agent.grad_norms.mean())
    sd1 = agent.policy_network.state_dict()
    for k in ("obs_embedding.observation_embedding.0.weight", "obs_embedding.observation_embedding.8.bias", "obs_embedding.observation_embedding.11.weight"):
        assert not torch.equal(sd0[k], sd1[k]), k
    # the replay holds the pixels the env produced
    arrays = agent.replay_buffer.export_arrays()
    assert arrays["obss"].dtype == np.uint8 and arrays["obss"].max() >= 200


def test_full_size_minihack_crop_update_properties(lib):
    """3 x 144 x 144 pixel windows, context 8, batch 4 (36 + 32 encoder tokens): determinism of the update, and batch independence
    of Q (a sequence's Q-values do not depend on what else is in the batch)."""
    import sys, os
    from oracle import dtqn_oracle as O
    from helpers import pack_theta
    from test_image_golden import image_net
    from dtqn_amd.learner import DeviceReplay, TdEngine
    cfg = O.NetCfg(obs_dim=3 * 144 * 144, num_actions=8, inner_embed_size=64, num_heads=8, num_layers=2, history_len=8, image=(3, 144, 144))
    pol, tgt = O.init_params(cfg, seed=9, perturb=True), O.init_params(cfg, seed=10, perturb=True)
    rng = np.random.default_rng(4)
    E, T, L = 6, 12, 8
    rows = rng.integers(0, 256, size=(E, T + 1, 3 * 144 * 144), dtype=np.uint8)

    def run(batch, eps, starts):
        net = image_net(lib, cfg)
        eng = TdEngine(net, batch)
        eng.theta_pol.copy_(torch.from_numpy(pack_theta(eng.net, pol))); eng.theta_tgt.copy_(torch.from_numpy(pack_theta(eng.net, tgt)))
        rep = DeviceReplay(E, T, cfg.image, 0, eng.device)
        rep.obs.copy_(torch.from_numpy(rows))
        rep.actions.copy_(torch.from_numpy(rng0.integers(0, 8, size=(E, T + 1)).astype(np.uint8)))
        rep.rewards.copy_(torch.from_numpy(rng0.choice(np.array([0, 1, -1], dtype=np.float32), size=(E, T))))
        rep.dones.zero_(); rep.ep_len.fill_(T)
        eng.set_indices(np.asarray(eps, dtype=np.int32), np.asarray(starts, dtype=np.int32))
        eng.forward_backward(rep)
        torch.cuda.synchronize()
        return eng.q3.cpu().numpy().reshape(3, batch, eng.net.lp, eng.net.ap)[:, :, :L, :8].copy(), eng.grad.cpu().numpy().copy()
    rng0 = np.random.default_rng(5)
    q4, g4 = run(4, [0, 1, 2, 3], [0, 1, 2, 3])
    rng0 = np.random.default_rng(5)
    q4b, g4b = run(4, [0, 1, 2, 3], [0, 1, 2, 3])
    assert np.array_equal(q4, q4b) and np.array_equal(g4, g4b)                       # deterministic
    rng0 = np.random.default_rng(5)
    q2, _ = run(2, [2, 3], [2, 3])
    assert np.isfinite(q4).all() and np.isfinite(g4).all() and np.abs(g4).max() > 0
    assert np.abs(q2 - q4[:, 2:]).max() <= 1e-5 * max(1.0, np.abs(q4).max())         # batch independence
    # and a token against the oracle (one window: the CPU convolutions take seconds at this size)
    with torch.no_grad():
        ref = O.forward(pol, cfg, torch.as_tensor(rows[0, 0:L].reshape(1, L, 3, 144, 144))).numpy()[0]
    assert np.abs(q4[0, 0] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_image_replay_producer_is_bit_exact(lib):
    """uint8 pixel rows through observe() / context_reset() -> pinned staging -> dtqn_replay_push -> HBM, against the oracle buffer fed
    the same calls (dtqn/buffers/replay_buffer.py:71-135 with the image array of :41-46): every byte of every slot, after the ring has
    wrapped, with records of several episodes (and the slot cleanses between them) inside one staged batch."""
    import run as runpy
    from oracle import replay_oracle as RO
    from dtqn_amd.utils.agent_utils import get_agent
    from dtqn_amd.utils.random import set_global_seed
    env = _PixelEnv(1)
    set_global_seed(5, env)
    T = env._max_episode_steps
    agent = get_agent("DTQN", [env], 8, 0, 64, 10 * T, torch.device("cuda"), 3e-4, 4, 8, -1, 8, 1000, 0.99, 4, 1, 0.0, False, "res", "learned", 0,
                      sampler="device", sample_seed=3)
    rb = agent.replay_buffer
    O_ = int(np.prod(agent.env_obs_length))
    shadow = RO.ReplayOracle(rb.max_size * T, O_, agent.obs_mask, T, agent.context_len)
    orig = (rb.store_obs, rb.store, rb.flush)
    flat = lambda o: np.asarray(o).reshape(-1)
    rb.store_obs = lambda o: (orig[0](o), shadow.store_obs(flat(o)))
    rb.store = lambda o, a, r, d, n=0: (orig[1](o, a, r, d, n), shadow.store(flat(o), a, r, d, n))
    rb.flush = lambda: (orig[2](), shadow.flush())
    runpy.prepopulate(agent, 40 * T, [env])
    assert rb.pos[0] > rb.max_size                      # the ring wrapped: slots were cleansed and rewritten on the device
    arrays = rb.export_arrays()
    assert arrays["obss"].dtype == np.uint8
    assert np.array_equal(arrays["obss"].reshape(shadow.obss.shape), shadow.obss.astype(np.uint8))
    assert np.array_equal(arrays["actions"], shadow.actions[:, :, 0])
    assert np.array_equal(arrays["rewards"], shadow.rewards[:, :, 0])
    assert np.array_equal(arrays["dones"].astype(bool), shadow.dones[:, :, 0])
    assert np.array_equal(rb.dev.ep_len.cpu().numpy(), shadow.episode_lengths) and list(rb.pos) == list(shadow.pos)
