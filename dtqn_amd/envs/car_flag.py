"""Car Flag: a 1-D car must reach the good flag; which side is good is only revealed near the
"priest" at x = 0.5 (reference dynamics: envs/car_flag.py:18-159, rendering dropped).

Host-side rollout env: plain numpy on the CPU cores, as north_star prescribes.
"""
import numpy as np
from numpy.random import Generator

from . import spaces

MAX_POSITION, MAX_SPEED, POWER = 1.1, 0.07, 0.0015
PRIEST_POSITION, PRIEST_DELTA = 0.5, 0.2


class CarFlag:
    def __init__(self, discrete: bool = True):
        self.discrete = discrete
        self.heaven_position, self.hell_position = 1.0, -1.0
        low = np.array([-MAX_POSITION, -MAX_SPEED, -1.0], dtype=np.float32)
        self.observation_space = spaces.Box(low=low, high=-low, shape=(3,), dtype=np.float32)
        self.action_space = spaces.Discrete(3) if discrete else spaces.Box(low=[-1.0], high=[1.0], shape=(1,))
        self.np_random = None
        self.state = None

    def seed(self, seed=None):
        # seeded once: later calls keep the stream (envs/car_flag.py:70-74)
        if self.np_random is None:
            self.np_random = Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return [seed]

    def reset(self):
        # draw order matters for stream parity: side first, then the start position
        side_is_right = self.np_random.integers(low=0, high=2, size=1) == 0
        self.heaven_position = 1.0 if side_is_right else -1.0
        self.hell_position = -self.heaven_position
        self.state = np.array([self.np_random.uniform(low=-0.2, high=0.2), 0, 0.0])
        return np.array(self.state)

    def step(self, action):
        position, velocity = self.state[0], self.state[1]
        force = (action - 1) if self.discrete else np.clip(action, -1, 1)     # 0,1,2 -> push left, coast, push right
        velocity = min(max(velocity + force * POWER, -MAX_SPEED), MAX_SPEED)
        position = min(max(position + velocity, -MAX_POSITION), MAX_POSITION)
        if position == -MAX_POSITION and velocity < 0:
            velocity = 0
        done = bool(position >= 1.0 or position <= -1.0)
        reward = 0
        if self.heaven_position > self.hell_position:
            if position >= self.heaven_position:
                reward = 1.0
            if position <= self.hell_position:
                reward = -1.0
        else:
            if position <= self.heaven_position:
                reward = 1.0
            if position >= self.hell_position:
                reward = -1.0
        direction = 0.0
        if PRIEST_POSITION - PRIEST_DELTA <= position <= PRIEST_POSITION + PRIEST_DELTA:
            direction = 1.0 if self.heaven_position > self.hell_position else -1.0
        self.state = np.array([position, velocity, direction])
        return self.state, reward, done, {"is_success": reward > 0}
