"""Vectorised rollout: N host environments per learner, ONE batched actor launch per vector step.

The reference steps one environment per network forward (run.py:356-377); at 1 env step : 1 update the actor forward
and the Python env step sit on the critical path of every update.  `VectorActor` keeps N environments and N rolling
contexts, stages all N prefixes through one pinned buffer, runs `dtqn_actor_forward_batch` (ragged prefixes in one
launch: every sequence runs max n_i rows, causality keeps shorter prefixes exact) and reads the N Q-rows back from
pinned memory.

Replay semantics stay the reference's: an episode becomes sampleable when it has FINISHED (replay_buffer.py:141-145
excludes the slot in progress).  With N episodes in progress at once, each environment collects its episode on the
host and replays it into the buffer's producer API (store_obs, store x len, flush) when it ends, so the device arrays
hold exactly what the single-environment loop would have written for that episode.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import numpy as np
import torch

from .. import _binding as B
from ..utils.bag import Bag
from ..utils.context import Context
from ..utils.random import RNG


class VectorActor:
    def __init__(self, agent, envs: Sequence, ref_quirks: bool = False):
        if getattr(agent, "image", None) is not None:
            # pixel observations act through DTQN.forward (convolutional embedding + row-block forward, one environment at a time);
            # the float32 staging and dtqn_actor_forward_batch below have no image path
            raise NotImplementedError("vectorised rollout of image observations (use --num-envs 1)")
        self.agent, self.envs = agent, list(envs)
        N = self.n = len(self.envs)
        L, O, A = agent.context_len, agent.env_obs_length, agent.num_actions
        self.L, self.O, self.A = L, O, A
        self.contexts: List[Context] = [Context(L, agent.obs_mask, A, O, discrete=agent.is_discrete_env, ref_quirks=ref_quirks)
                                        for _ in range(N)]
        # bag networks: one bag per environment (utils/bag.py; dtqn.py:66-74 keeps one per agent because it steps one environment)
        self.bags = [Bag(agent.bag.size, agent.obs_mask, O, discrete=agent.is_discrete_env, ref_quirks=ref_quirks)
                     for _ in range(N)] if agent.bag.size > 0 else None
        self.episodes = [[] for _ in range(N)]            # per env: [first_obs, (obs, action, reward, done), ...]
        self.returns = np.zeros(N)
        cuda = agent.device.type == "cuda"
        obs_bytes = N * L * O * 4
        act_bytes = (N * L + 3) & ~3
        total = obs_bytes + act_bytes + 4 * N
        pin = (lambda t: t.pin_memory()) if cuda else (lambda t: t)
        self._ctx_h = pin(torch.zeros(total, dtype=torch.uint8))
        self._ctx_d = torch.zeros(total, dtype=torch.uint8, device=agent.device)
        buf = self._ctx_h.numpy()
        self._obs_np = buf[:obs_bytes].view(np.float32).reshape(N, L, O)
        self._act_np = buf[obs_bytes:obs_bytes + N * L].reshape(N, L)
        self._len_np = buf[obs_bytes + act_bytes:].view(np.int32)
        self._q_d = torch.zeros(N * L * A, device=agent.device)
        self._q_h = pin(torch.zeros(N, A))
        self._q_np = self._q_h.numpy()
        eng = agent.engine
        need = eng.lib.dtqn_forward_workspace_floats(eng._actor_net_ref, N)
        self._ws = torch.zeros(max(1, need), dtype=torch.float32, device=agent.device)
        self._ws_p = ctypes.c_void_p(self._ws.data_ptr()) if need > 0 else None
        self._p = [ctypes.c_void_p(t.data_ptr()) for t in (self._ctx_h, self._ctx_d, self._q_d, self._q_h)]
        self._ev = torch.cuda.Event() if cuda else None      # completion of the batched actor forward alone
        self.steps = 0
        self.episodes_done = 0

    # ------------------------------------------------------------------------------------------
    def reset_all(self) -> None:
        for i, env in enumerate(self.envs):
            self._reset(i)

    def _reset(self, i: int) -> None:
        obs = self.envs[i].reset()
        self.contexts[i].reset(obs)
        if self.bags is not None:
            self.bags[i].reset()
        self.episodes[i] = [np.array(obs, copy=True)]
        self.returns[i] = 0.0

    def _launch_q(self) -> None:
        """Stage all N contexts and launch the batched actor forward on the learner's stream (no synchronisation)."""
        a, eng = self.agent, self.agent.engine
        if self.bags is not None:
            # bag networks: the module forward with the N bags (dtqn_forward_bag); every sequence runs the longest prefix, the
            # rows behind a shorter one cannot reach its last live row (causal), its own bag attends row by row
            lens = [min(c.max_length, c.timestep + 1) for c in self.contexts]
            groups = [list(range(self.n))]
            if a.policy_network.net.action_dim > 0 and max(lens) > 1 and min(lens) == 1:
                # a ONE-row sequence keeps its action embedding un-rolled (dtqn.py:187-191: `if history_len > 1`); run next to longer
                # prefixes it would be rolled and zeroed, so the fresh episodes of this vector step get a forward of their own
                groups = [[i for i in range(self.n) if lens[i] > 1], [i for i in range(self.n) if lens[i] == 1]]
            q_rows = None
            for idx in groups:
                n_max = max(lens[i] for i in idx)
                obs = np.stack([self.contexts[i].obs[:n_max] for i in idx])
                act = np.stack([self.contexts[i].action[:n_max] for i in idx])
                q = a._bag_forward(obs, act, np.stack([self.bags[i].obss for i in idx]), np.stack([self.bags[i].actions for i in idx]))
                rows = torch.as_tensor(np.asarray([lens[i] for i in idx]) - 1, device=q.device)
                last = q[torch.arange(len(idx), device=q.device), rows]           # the last live row of every sequence
                if q_rows is None:
                    q_rows = torch.empty(self.n, self.A, dtype=q.dtype, device=q.device)
                q_rows[torch.as_tensor(idx, device=q.device)] = last
            # -> pinned memory with one asynchronous copy, and an event right behind it: updates queued after this point
            # (step_all's `between`) no longer stand between the host and its Q-values
            if self._ev is not None:
                self._q_h.copy_(q_rows, non_blocking=True)
                self._ev.record(torch.cuda.current_stream(a.device))
            else:
                self._q_h.copy_(q_rows)
            return
        n_max = 1
        for i, ctx in enumerate(self.contexts):
            n = min(ctx.max_length, ctx.timestep + 1)
            self._obs_np[i, :n] = ctx.obs[:n]
            self._act_np[i, :n] = ctx.action[:n, 0]
            self._len_np[i] = n
            n_max = max(n_max, n)
        a._actor_calls += 1
        rc = eng.lib.dtqn_actor_forward_batch(eng._actor_net_ref, a._theta_p, self._p[0], self._p[1], self.n, n_max, self._p[2], self._p[3],
                                              self._ws_p, 1 if a.train_mode.name == "TRAIN" else 0, eng.td.dropout_seed ^ 0xAC70, a._actor_calls & 0xFFFFFFFF, eng._stream())
        if rc == B.DEFINES["DTQN_ERR_ARG"]:
            raise AssertionError("Cannot forward, history is longer than expected.")   # dtqn.py:170-173
        if rc != 0:
            raise RuntimeError(f"dtqn_actor_forward_batch failed with DTQN status {rc}")
        if self._ev is not None:
            self._ev.record(a._main_stream)

    def _wait_q(self) -> np.ndarray:
        if self._ev is not None:
            self._ev.synchronize()              # the forward only: work queued behind it (TD updates) keeps running
        return self._q_np

    def q_values(self) -> np.ndarray:
        """Q[:, -1] of every actor's current context: one launch, [N][A] (pinned host view; valid until the next call)."""
        self._launch_q()
        return self._wait_q()

    def act(self, epsilon: float, between=None) -> np.ndarray:
        """Epsilon-greedy actions for all N environments (dtqn.py:76-107 per actor; draws from RNG.rng in env order).
        between(): called after the actor forward has been launched and before its result is awaited -- the place to
        queue GPU work that may run while the host steps the environments."""
        explore = RNG.rng.random(self.n) < epsilon
        actions = np.zeros(self.n, dtype=np.int64)
        greedy = not explore.all()
        if greedy:
            self._launch_q()
        if between is not None:
            between()
        if greedy:
            actions[:] = np.argmax(self._wait_q(), axis=1)            # first max, like torch.argmax
        if explore.any():
            actions[explore] = RNG.rng.integers(self.A, size=int(explore.sum()))
        return actions

    def step_all(self, epsilon: float, updates: int = 0) -> int:
        """One vector step: act, step every environment, record; finished episodes are replayed into the buffer and their
        environments reset.  updates > 0: that many agent.train() calls are QUEUED right behind the actor forward, so the
        GPU runs them while the host steps the N environments (the actions of this vector step come from the parameters
        before those updates, exactly as when train() is called after the step).  Returns the number of finished episodes."""
        agent = self.agent

        def queue_updates():
            for _ in range(updates):
                agent.train()
        actions = self.act(epsilon, queue_updates if updates > 0 else None)
        done_count = 0
        for i, env in enumerate(self.envs):
            a = int(actions[i])
            obs, reward, done, info = env.step(a)
            stored_done = False if info.get("TimeLimit.truncated", False) else done     # run.py:368-376
            evicted_obs, evicted_action = self.contexts[i].add_transition(obs, a, reward, stored_done)
            if self.bags is not None and evicted_obs is not None:
                agent._bag_insert(self.bags[i], self.contexts[i], evicted_obs, evicted_action)
            self.episodes[i].append((np.array(obs, copy=True), a, float(reward), bool(stored_done)))
            self.returns[i] += reward
            if done:
                self._commit_episode(i)
                self._reset(i)
                done_count += 1
        self.steps += self.n
        self.episodes_done += done_count
        return done_count

    def _commit_episode(self, i: int) -> None:
        rb = self.agent.replay_buffer
        ep = self.episodes[i]
        rb.store_obs(ep[0])
        for t, (obs, a, r, d) in enumerate(ep[1:]):
            rb.store(obs, a, r, d, t + 1)
        rb.flush()
