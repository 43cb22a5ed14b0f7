"""Agent / network factory with the reference's signature (utils/agent_utils.py:36-168)."""
from typing import Sequence

import numpy as np
import torch

from ..agents.dtqn import DtqnAgent
from ..networks.dtqn import DTQN
from . import env_processing

# The reference also registers LSTM / MLP baselines (ADRQN, DRQN, DARQN, DQN) here; they are a
# different model family and outside this engine's scope.
MODEL_MAP = {"DTQN": DTQN}
AGENT_MAP = {"DTQN": DtqnAgent}


def get_agent(model_str: str, envs: Sequence, embed_per_obs_dim: int, action_dim: int, inner_embed: int,
              buffer_size: int, device, learning_rate: float, batch_size: int, context_len: int, max_env_steps: int,
              history: int, target_update_frequency: int, gamma: float, num_heads: int = 1, num_layers: int = 1,
              dropout: float = 0.0, identity: bool = False, gate: str = "res", pos: str = "learned", bag_size: int = 0,
              **agent_kwargs):
    """Build the agent (policy + target network, replay buffer, contexts) for `envs`, which must
    share observation and action spaces.  Positional order = the reference's, so run.py-style
    call sites work unchanged."""
    if model_str not in MODEL_MAP:
        raise NotImplementedError(f"model {model_str!r}: dtqn_amd implements {sorted(MODEL_MAP)} (DTQN hot path only)")
    env0 = envs[0]
    obs_len = env_processing.get_env_obs_length(env0)
    obs_mask = env_processing.get_env_obs_mask(env0)
    if max_env_steps <= 0:
        max_env_steps = max(env_processing.get_env_max_steps(env) for env in envs)
    vocab = int(np.max(obs_mask)) + 1
    discrete = env_processing.is_discrete_env(env0)
    if history < 1 or history > context_len:
        clipped = int(np.clip(history, 1, context_len))
        print(f"History must be 1 < history <= context_len, but history is {history} and context len is "
              f"{context_len}. Clipping history to {clipped}...")
        history = clipped
    num_actions = env0.action_space.n
    device = torch.device(device)

    def network_factory():
        return MODEL_MAP[model_str](
            obs_len, num_actions, embed_per_obs_dim, action_dim, inner_embed, num_heads, num_layers, context_len,
            dropout=dropout, gate=gate, identity=identity, pos=pos, discrete=discrete, vocab_sizes=vocab,
            target_update_frequency=target_update_frequency, bag_size=bag_size).to(device)

    return AGENT_MAP[model_str](
        network_factory, buffer_size, device, obs_len, max_env_steps, obs_mask, num_actions, discrete,
        learning_rate=learning_rate, batch_size=batch_size, gamma=gamma, context_len=context_len,
        embed_size=inner_embed, history=history, target_update_frequency=target_update_frequency, bag_size=bag_size,
        **agent_kwargs)
