"""Memory card game: N pairs face down, one card is shown per step and the agent must point at its
twin (reference dynamics: envs/memory_cards.py:8-116).  Observation: per card 0 = hidden,
1..N = face value, N+1 = removed."""
import numpy as np
from numpy.random import Generator

from . import spaces


class Memory:
    def __init__(self, num_pairs: int = 5):
        self.num_pairs, self.num_cards = num_pairs, 2 * num_pairs
        self.observation_space = spaces.MultiDiscrete([num_pairs + 2] * self.num_cards)
        self.action_space = spaces.Discrete(self.num_cards)
        self.card_removed, self.card_hidden = num_pairs + 1, 0
        self.state = self.card_removed * np.ones(self.num_cards)
        self.observation = self.card_hidden * np.ones(self.num_cards)
        self.current_card = -1
        self.np_random = None

    def seed(self, seed=None):
        if self.np_random is None:
            self.np_random = Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return [seed]

    def _all_removed(self, arr) -> bool:
        return bool(np.all(arr == self.card_removed))

    def reset(self):
        self.state = np.repeat(np.arange(1, self.num_pairs + 1), 2)
        self.np_random.shuffle(self.state)
        self.observation = self.card_hidden * np.ones(self.num_cards)
        self.current_card = self.np_random.integers(self.num_cards)
        self.observation[self.current_card] = self.state[self.current_card]
        return self.observation.copy()

    def step(self, action: int):
        if self._all_removed(self.state):
            raise ValueError("Trying to take step in invalid state. Did you reset?")
        done, info, cur = False, {}, self.current_card
        if action != cur and self.state[action] == self.observation[cur]:
            # found the twin: both cards leave the table (note: a card already removed still "matches"
            # when the shown value equals its hidden state -- the reference compares against state)
            self.observation[action] = self.card_removed
            self.observation[cur] = self.card_removed
            reward = 0
            if self._all_removed(self.observation):
                done = True
                info["is_success"] = True
        else:
            self.observation[cur] = self.card_hidden
            reward = -1
        if not done:
            self.current_card = self.np_random.integers(self.num_cards)
            while self.observation[self.current_card] == self.card_removed:
                self.current_card = self.np_random.integers(self.num_cards)
            self.observation[self.current_card] = self.state[self.current_card]
        return self.observation.copy(), reward, done, info
