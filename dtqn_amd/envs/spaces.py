"""Minimal observation / action space descriptors (gym is not a dependency of the hot path)."""
import numpy as np


class Space:
    def seed(self, seed=None):
        return [seed]


class Discrete(Space):
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = ()


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape


class MultiBinary(Space):
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = (self.n,)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype), dtype
        self.shape = tuple(shape) if shape is not None else self.low.shape
