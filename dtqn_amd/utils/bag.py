"""Persistent-memory bag: observation/action pairs that fell out of the agent's context (reference: utils/bag.py)."""
from typing import Optional, Tuple

import numpy as np


class Bag:
    """Fixed-size store of evicted (obs, action) pairs; `obss` [size, obs_length], `actions` [size, 1].

    `ref_quirks=True` reproduces the reference's storage dtype: its dtype-less np.full takes the dtype of the padding value,
    so the integer mask of continuous envs (-5) makes an int64 array that TRUNCATES float observations (utils/bag.py:42-51, the
    same quirk as the Context's).  The default stores float32 for continuous observations, int64 for discrete ones."""

    def __init__(self, bag_size: int, obs_mask, obs_length: int, discrete: Optional[bool] = None, ref_quirks: bool = False):
        if isinstance(obs_length, tuple) and bag_size > 0:
            raise NotImplementedError("a persistent-memory bag of image observations is outside the gfx950 kernels' coverage")
        self.size = bag_size
        self.obs_mask = obs_mask
        self.obs_length = obs_length
        self.discrete, self.ref_quirks = discrete, ref_quirks
        self.pos = 0
        self.obss, self.actions = self.make_empty_bag()

    def reset(self) -> None:
        self.pos = 0
        self.obss, self.actions = self.make_empty_bag()

    def add(self, obs: np.ndarray, action: int) -> bool:
        """Append while there is room; a full bag rejects the pair (the agent then decides what to evict)."""
        if self.is_full:
            return False
        self.obss[self.pos] = obs
        self.actions[self.pos] = action
        self.pos += 1
        return True

    def export(self) -> Tuple[np.ndarray, np.ndarray]:
        return self.obss[: self.pos], self.actions[: self.pos]

    def make_empty_bag(self) -> Tuple[np.ndarray, np.ndarray]:
        shape = (self.size, *self.obs_length) if isinstance(self.obs_length, tuple) else (self.size, self.obs_length)
        if self.ref_quirks or self.discrete is None:
            obss = np.full(shape, self.obs_mask)
        else:
            obss = np.full(shape, self.obs_mask, dtype=np.int64 if self.discrete else np.float32)
        return obss, np.full((self.size, 1), 0)

    @property
    def is_full(self) -> bool:
        return self.pos >= self.size
