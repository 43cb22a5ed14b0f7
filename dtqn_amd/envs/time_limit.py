"""Episode step cap with gym-0.18 TimeLimit semantics (the reference registers its envs with
max_episode_steps, envs/__init__.py:31-48, and run.py:369-374 reads info["TimeLimit.truncated"])."""


class TimeLimit:
    def __init__(self, env, max_episode_steps: int):
        self.env = env
        self._max_episode_steps = int(max_episode_steps)
        self._elapsed_steps = None
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    def seed(self, seed=None):
        return self.env.seed(seed)

    def reset(self):
        self._elapsed_steps = 0
        return self.env.reset()

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, reward, done, info

    def render(self, *a, **k):
        raise NotImplementedError("rendering is outside the scope of dtqn_amd")

    def __getattr__(self, name):
        return getattr(self.env, name)
