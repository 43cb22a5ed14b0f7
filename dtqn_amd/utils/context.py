"""Rolling actor-side history of the last `max_length` transitions (reference: utils/context.py)."""
from typing import Optional, Tuple

import numpy as np

from .random import RNG


class Context:
    """Window of (obs, action, reward, done) the actor feeds to the network.

    `ref_quirks=True` reproduces the reference's storage dtype: it builds the observation array
    with a dtype-less np.full, so with the integer padding value of continuous envs (-5) the array
    is int64 and float observations are TRUNCATED toward zero when written (utils/context.py:46-48,
    SURVEY.md section 4 quirk 2).  The default stores float32 for continuous observations."""

    def __init__(self, context_length: int, obs_mask, num_actions: int, env_obs_length: int,
                 init_hidden=None, discrete: Optional[bool] = None, ref_quirks: bool = False):
        self.max_length = context_length
        self.env_obs_length = env_obs_length
        self.num_actions = num_actions
        self.obs_mask = obs_mask
        self.reward_mask, self.done_mask = 0.0, True
        self.timestep = 0
        self.init_hidden = init_hidden
        self.ref_quirks = ref_quirks
        self.discrete = discrete

    def _obs_dtype(self):
        if self.ref_quirks:
            return None                       # whatever np.full infers from the mask, like the reference
        return np.int64 if self.discrete else np.float32

    def reset(self, obs: np.ndarray) -> None:
        dt = self._obs_dtype()
        if isinstance(self.env_obs_length, tuple):       # images: uint8 pixels (utils/context.py:41-47)
            self.obs = np.full([self.max_length, *self.env_obs_length], self.obs_mask, dtype=np.uint8)
        else:
            shape = [self.max_length, self.env_obs_length]
            self.obs = np.full(shape, self.obs_mask) if dt is None else np.full(shape, self.obs_mask, dtype=dt)
        self.obs[0] = obs
        # padding actions are random draws from the exploration stream (utils/context.py:50)
        self.action = RNG.rng.integers(self.num_actions, size=(self.max_length, 1))
        self.reward = np.full_like(self.action, self.reward_mask)
        self.done = np.full_like(self.reward, self.done_mask, dtype=np.int32)
        self.hidden = self.init_hidden
        self.timestep = 0

    @property
    def is_full(self) -> bool:
        return self.timestep >= self.max_length

    def add_transition(self, o, a: int, r: float, done: bool) -> Tuple[Optional[np.ndarray], Optional[int]]:
        """Append a transition, sliding the window once it is full.  Returns the evicted
        (obs, action) -- only a bag would consume it -- or (None, None)."""
        self.timestep += 1
        if self.is_full:
            # np.roll(x, -1, axis=0) of the reference (utils/context.py:60-64), in place: four fresh arrays per environment step
            # were 10 % of the actor loop's host time
            for arr in (self.obs, self.action, self.reward, self.done):
                first = arr[0].copy()
                arr[:-1] = arr[1:]
                arr[-1] = first
        slot = min(self.timestep, self.max_length - 1)
        evicted = (self.obs[slot].copy(), self.action[slot]) if self.is_full else (None, None)
        self.obs[slot] = o
        self.action[slot] = a
        self.reward[slot] = r
        self.done[slot] = done
        return evicted

    def export(self):
        cur = min(self.timestep, self.max_length) - 1
        return self.obs[cur + 1], self.action[cur], self.reward[cur], self.done[cur]

    @staticmethod
    def context_like(context: "Context") -> "Context":
        return Context(context.max_length, context.obs_mask, context.num_actions, context.env_obs_length,
                       init_hidden=context.init_hidden, discrete=context.discrete, ref_quirks=context.ref_quirks)
