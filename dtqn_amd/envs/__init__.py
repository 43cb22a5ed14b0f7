"""Host-side rollout environments and their registry (reference ids: envs/__init__.py:31-48).

Only the envs that are self-contained in the reference are restated (CarFlag, Memory).  The
gridverse / gym-pomdps / MiniHack families need third-party simulators that are not vendored by the
reference and are absent here; `make` falls through to `gym.make` for them when gym is installed.
"""
from .car_flag import CarFlag
from .memory_cards import Memory
from .time_limit import TimeLimit

REGISTRY = {
    "DiscreteCarFlag-v0": lambda: TimeLimit(CarFlag(discrete=True), max_episode_steps=200),
    "Memory-5-v0": lambda: TimeLimit(Memory(num_pairs=5), max_episode_steps=50),
}


def make(env_id: str):
    if env_id in REGISTRY:
        return REGISTRY[env_id]()
    try:
        import gym
    except ImportError as exc:
        raise KeyError(
            f"Unknown environment {env_id!r}: dtqn_amd ships {sorted(REGISTRY)}; gridverse / gym-pomdps / MiniHack "
            "domains need their third-party packages plus gym, which are not installed") from exc
    return gym.make(env_id)
