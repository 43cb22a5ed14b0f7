"""CPU oracle for the replay path.  TEST INFRASTRUCTURE ONLY (see oracle/dtqn_oracle.py header).

Plain-numpy restatement of the reference's episode-major replay buffer and rolling context
(dtqn/buffers/replay_buffer.py, utils/context.py, utils/epsilon_anneal.py, utils/logging_utils.py),
pinned by tests/golden/G5_replay.npz and G7_misc.npz (tests/test_host_golden.py).  One deliberate
deviation, applied by the golden harness to the reference too: episode_lengths is int64 instead of
uint8 (numpy >= 2 wraps `uint8 - 50`, replay_buffer.py:69,152).
"""
from __future__ import annotations

import random
from typing import Tuple

import numpy as np


class ReplayOracle:
    """replay_buffer.py:19-168 (1-D observations; no bag)."""

    def __init__(self, buffer_size: int, env_obs_length: int, obs_mask, max_episode_steps: int, context_len: int = 1):
        self.max_size = buffer_size // max_episode_steps                       # :27
        self.context_len = context_len
        self.env_obs_length = env_obs_length
        self.max_episode_steps = max_episode_steps
        self.obs_mask = obs_mask
        self.pos = [0, 0]
        T, E = max_episode_steps, self.max_size
        self.obss = np.full([E, T + 1, env_obs_length], obs_mask, dtype=np.float32)   # :47-55
        self.actions = np.zeros([E, T + 1, 1], dtype=np.uint8)                        # :58-61
        self.rewards = np.zeros([E, T, 1], dtype=np.float32)
        self.dones = np.ones([E, T, 1], dtype=np.bool_)
        self.episode_lengths = np.zeros([E], dtype=np.int64)

    def store(self, obs, action, reward, done, episode_length=0) -> None:          # :71-86
        e, t = self.pos[0] % self.max_size, self.pos[1]
        self.obss[e, t + 1] = obs
        self.actions[e, t] = action
        self.rewards[e, t] = reward
        self.dones[e, t] = done
        self.episode_lengths[e] = episode_length
        self.pos = [self.pos[0], self.pos[1] + 1]

    def store_obs(self, obs) -> None:                                              # :88-92
        e = self.pos[0] % self.max_size
        self.cleanse_episode(e)
        self.obss[e, 0] = obs

    def can_sample(self, batch_size: int) -> bool:                                 # :94-95
        return batch_size < self.pos[0]

    def flush(self) -> None:                                                       # :97-98
        self.pos = [self.pos[0] + 1, 0]

    def cleanse_episode(self, e: int) -> None:                                     # :100-135
        self.obss[e] = self.obs_mask
        self.actions[e] = 0
        self.rewards[e] = 0.0
        self.dones[e] = True
        self.episode_lengths[e] = 0

    def sample_indices(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray]:
        """The index draw of sample() (:141-158): consumes Python's `random` exactly like the reference."""
        valid = [i for i in range(min(self.pos[0], self.max_size)) if i != self.pos[0] % self.max_size]
        eps = np.array([random.choice(valid) for _ in range(batch_size)])
        starts = np.array([random.randint(0, max(0, int(self.episode_lengths[e]) - self.context_len)) for e in eps])
        return eps, starts

    def gather(self, eps: np.ndarray, starts: np.ndarray):
        """The fancy-index gather of sample() (:156-167)."""
        e = np.asarray(eps).reshape(-1, 1)
        tr = np.asarray(starts).reshape(-1, 1) + np.arange(self.context_len)[None, :]
        return (self.obss[e, tr], self.actions[e, tr], self.rewards[e, tr], self.obss[e, 1 + tr],
                self.actions[e, 1 + tr], self.dones[e, tr],
                np.clip(self.episode_lengths[e], 0, self.context_len))

    def sample(self, batch_size: int):
        return self.gather(*self.sample_indices(batch_size))


def synth_fill(buf, rng: np.random.Generator, n_eps: int, discrete: bool, vocab: int, num_actions: int, min_len: int = 3):
    """Synthetic replay content of SURVEY.md section 8d through the buffer's own producer API."""
    T = buf.max_episode_steps
    for _ in range(n_eps):
        n = int(rng.integers(min_len, T + 1))
        if discrete:
            obs = rng.integers(0, max(1, vocab - 1), size=(n + 1, buf.env_obs_length)).astype(np.float32)
        else:
            obs = rng.uniform(-1, 1, size=(n + 1, buf.env_obs_length)).astype(np.float32)
        act = rng.integers(0, num_actions, size=n)
        rew = rng.choice(np.array([0, 0, 0, 1, -1], dtype=np.float32), size=n)
        buf.store_obs(obs[0])
        for t in range(n):
            buf.store(obs[t + 1], int(act[t]), float(rew[t]), bool(t == n - 1), t + 1)
        buf.flush()


class ContextOracle:
    """utils/context.py:19-96 (1-D observations).  `truncate=True` reproduces the reference's
    dtype-less np.full (int64 storage with an integer mask -> float observations truncated toward
    zero)."""

    def __init__(self, context_length: int, obs_mask, num_actions: int, env_obs_length: int, rng, truncate: bool = True):
        self.max_length, self.obs_mask, self.num_actions, self.env_obs_length = context_length, obs_mask, num_actions, env_obs_length
        self.rng, self.truncate, self.timestep = rng, truncate, 0

    def reset(self, obs) -> None:                                                  # :36-54
        if self.truncate:
            self.obs = np.full([self.max_length, self.env_obs_length], self.obs_mask)
        else:
            self.obs = np.full([self.max_length, self.env_obs_length], self.obs_mask, dtype=np.float32)
        self.obs[0] = obs
        self.action = self.rng.integers(self.num_actions, size=(self.max_length, 1))
        self.reward = np.full_like(self.action, 0.0)
        self.done = np.full_like(self.reward, True, dtype=np.int32)
        self.timestep = 0

    def add_transition(self, o, a, r, done) -> None:                               # :56-80
        self.timestep += 1
        if self.timestep >= self.max_length:
            self.obs, self.action = np.roll(self.obs, -1, axis=0), np.roll(self.action, -1, axis=0)
            self.reward, self.done = np.roll(self.reward, -1, axis=0), np.roll(self.done, -1, axis=0)
        t = min(self.timestep, self.max_length - 1)
        self.obs[t] = o
        self.action[t] = np.array([a])
        self.reward[t] = np.array([r])
        self.done[t] = np.array([done])


def linear_anneal_trace(start: float, end: float, duration: int, n: int) -> np.ndarray:
    """utils/epsilon_anneal.py:28-34: val <- max(end, val - (val - end)/duration)."""
    out, val = [], start
    for _ in range(n):
        out.append(val)
        val = max(end, val - (val - end) / duration)
    return np.array(out)
