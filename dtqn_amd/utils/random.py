"""Seeding and the global exploration RNG (reference: utils/random.py:9-31).

Two independent streams exist in the reference and are kept: Python's `random` drives replay
sampling, a numpy PCG64 generator (`RNG.rng`) drives epsilon-greedy and context initialisation.
"""
import os
import random

import numpy as np
import torch


class RNG:
    rng: np.random.Generator = None


def set_global_seed(seed: int, *envs) -> None:
    """Seed `random`, torch, numpy, every env (and its spaces) and RNG.rng, in the reference's order
    so that the three derived seeds (torch, numpy, hash) are the same numbers."""
    random.seed(seed)
    torch_seed = random.randint(1, int(1e6))
    numpy_seed = random.randint(1, int(1e6))
    hash_seed = random.randint(1, int(1e6))
    torch.manual_seed(torch_seed)
    np.random.seed(numpy_seed)
    for env in envs:
        env.seed(seed=seed)
        env.observation_space.seed(seed=seed)
        env.action_space.seed(seed=seed)
    os.environ["PYTHONHASHSEED"] = str(hash_seed)
    RNG.rng = np.random.Generator(np.random.PCG64(seed=seed))
