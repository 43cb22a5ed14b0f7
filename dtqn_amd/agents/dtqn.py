"""DTQN agent (actor + learner) with the reference's surface (dtqn/agents/dtqn.py:15-269 and its
base class dtqn/agents/dqn.py:24-327), running on the gfx950 engine.

What changed underneath:
  * `train()` launches the fused HIP TD update (five kernels, dtqn_amd.learner.TdEngine) on the
    device-resident replay; nothing of the batch ever exists on the host.
  * the seven logged statistics come back through a pinned ring, asynchronously; the reference's
    nine blocking `.item()` calls per update are gone.  RunningAverage.mean() drains the ring.
  * `get_action()` stages the rolling context through pinned memory, one H2D + one D2H per env step.
  * with torch.distributed initialised (one process per GPU, RCCL) the flat gradient is all-reduced
    between the gradient and the optimizer kernels (dtqn_amd.dist).
"""
from __future__ import annotations

import ctypes
import os
import random
from enum import Enum
from typing import Callable, Optional, Tuple, Union

import numpy as np
import torch

from .. import dist as ddp
from ..buffers.replay_buffer import ReplayBuffer
from .. import _binding as B
from ..learner import STAT_NAMES, TdEngine
from ..utils.bag import Bag
from ..utils.context import Context
from ..utils.logging_utils import DeferredRunningAverage, RunningAverage
from ..utils.random import RNG


# text of torch.nn.utils.clip_grad_norm_'s exception (norm_type 2.0), which dtqn/agents/dtqn.py:257-261 lets escape; pinned to the
# reference's own run by tests/golden G12 (`nonfinite/message`)
NONFINITE_MESSAGE = ("The total norm of order 2.0 for gradients from `parameters` is non-finite, so it cannot be clipped. To disable this error "
                     "and scale the gradients by the non-finite norm anyway, set `error_if_nonfinite=False`")


class TrainMode(Enum):
    TRAIN = 1
    EVAL = 2


class _AdamHandle:
    """Stand-in for `agent.optimizer` (dqn.py:64): the Adam state lives in the engine's flat m / v
    buffers; this exposes it for checkpoints in plain-array form."""

    def __init__(self, eng: TdEngine, before_load: Optional[Callable[[], None]] = None):
        self._eng = eng
        self._before_load = before_load            # the agent's side of a reload: its statistics ring restarts with the device's counter

    def state_dict(self) -> dict:
        e = self._eng
        return {"step": int(e.step_counter[1].item()), "exp_avg": e.adam_m.cpu().numpy(), "exp_avg_sq": e.adam_v.cpu().numpy(),
                "lr": e.td.lr, "betas": (e.td.beta1, e.td.beta2), "eps": e.td.eps}

    def load_state_dict(self, sd: dict) -> None:
        e = self._eng
        if self._before_load is not None:
            self._before_load()
        e.adam_m.copy_(torch.as_tensor(sd["exp_avg"])); e.adam_v.copy_(torch.as_tensor(sd["exp_avg_sq"]))
        e.step_counter[:2] = int(sd["step"])     # optimizer steps: published | next
        e.step_counter[2:] = 0         # statistics-ring call counter (DtqnAgent.load_checkpoint restarts its host side too) | skip flag
        e.pipeline_reset()


class DtqnAgent:
    # updates whose statistics may sit in the pinned ring before the host folds them into the running averages (a non-finite
    # gradient norm is therefore raised at most this many updates late; readers of the averages drain everything first)
    STATS_DRAIN_EVERY = 16

    def __init__(self, network_factory: Callable[[], torch.nn.Module], buffer_size: int, device: torch.device,
                 env_obs_length: int, max_env_steps: int, obs_mask: Union[int, float], num_actions: int,
                 is_discrete_env: bool, learning_rate: float = 0.0003, batch_size: int = 32, context_len: int = 50,
                 gamma: float = 0.99, grad_norm_clip: float = 1.0, target_update_frequency: int = 10_000,
                 history: int = 50, bag_size: int = 0, sampler: str = "reference", ref_quirks: bool = False,
                 sample_seed: int = 0, data_parallel: bool = True, **kwargs):
        self.context_len, self.env_obs_length = context_len, env_obs_length
        self.image = tuple(env_obs_length) if isinstance(env_obs_length, (tuple, list)) else None     # (C, H, W) pixel observations
        self.device = torch.device(device)
        self.policy_network = network_factory()
        self.target_network = network_factory()
        lib = self.policy_network._lib
        self._test_mode = self.device.type != "cuda"          # CPU kernel-emulation tests only
        self.num_actions, self.obs_mask, self.history = num_actions, obs_mask, history
        self.batch_size, self.gamma = batch_size, gamma
        self.grad_norm_clip, self.target_update_frequency = grad_norm_clip, target_update_frequency
        self.is_discrete_env = is_discrete_env
        self.obs_context_type = np.int_ if is_discrete_env else np.float32
        self.obs_tensor_type = torch.long if is_discrete_env else torch.float32
        self.sampler, self.sample_seed = sampler, int(sample_seed)
        self._separate_sample_launch = os.environ.get("DTQN_SAMPLE_LAUNCH", "0") == "1"
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' (Python `random` stream) or 'device'")
        self.engine = TdEngine(self.policy_network.net, batch_size, lr=learning_rate, gamma=gamma, history=history,
                               tuf=target_update_frequency, grad_norm_clip=grad_norm_clip, dropout_seed=int(sample_seed) + 0x5EED,
                               device=None if self._test_mode else self.device,
                               _test_lib=lib if self._test_mode else None,
                               theta_pol=self.policy_network.flat, theta_tgt=self.target_network.flat)
        self.target_update()
        self.target_network.eval()
        self.optimizer = _AdamHandle(self.engine, self._restart_stats_ring)
        self.replay_buffer = ReplayBuffer(buffer_size, env_obs_length=env_obs_length, obs_mask=obs_mask,
                                          max_episode_steps=max_env_steps, context_len=context_len, device=self.device, lib=lib)
        # one process per GPU under torch.distributed: the gradient exchange joins the update (data_parallel=False keeps this agent a
        # solo learner inside a distributed job -- bench.py measures its own one-GPU rate that way)
        self.dp = ddp.DataParallel(self.engine) if (data_parallel and ddp.is_distributed()) else None
        self._dp_ready = False
        if self.dp is not None:
            self.dp.broadcast_parameters()
        # logging (dqn.py:80-89), fed asynchronously
        self.num_train_steps = 0
        drain = lambda: self._drain_stats(block=True)
        self.td_errors, self.grad_norms = DeferredRunningAverage(100, drain), DeferredRunningAverage(100, drain)
        self.qvalue_max, self.target_max = DeferredRunningAverage(100, drain), DeferredRunningAverage(100, drain)
        self.qvalue_mean, self.target_mean = DeferredRunningAverage(100, drain), DeferredRunningAverage(100, drain)
        self.qvalue_min, self.target_min = DeferredRunningAverage(100, drain), DeferredRunningAverage(100, drain)
        self._stat_sinks = {"td_error": self.td_errors, "grad_norm": self.grad_norms, "qvalue_max": self.qvalue_max,
                            "qvalue_mean": self.qvalue_mean, "qvalue_min": self.qvalue_min, "target_max": self.target_max,
                            "target_mean": self.target_mean, "target_min": self.target_min}
        self._stat_index = (STAT_NAMES.index("nonfinite"), [(name, STAT_NAMES.index(name)) for name in self._stat_sinks])
        cuda = self.device.type == "cuda"
        self._calls_issued = 0        # dtqn_td_clip_adam launches so far
        self._calls_read = 0          # ... whose statistics have been consumed from the pinned ring
        self.train_mode = TrainMode.TRAIN
        mk = lambda: Context(context_len, obs_mask, num_actions, env_obs_length, discrete=is_discrete_env, ref_quirks=ref_quirks)
        self.train_context, self.eval_context = mk(), mk()
        mkb = lambda: Bag(bag_size, obs_mask, env_obs_length, discrete=is_discrete_env, ref_quirks=ref_quirks)
        self.train_bag, self.eval_bag = mkb(), mkb()                   # dtqn.py:66-67
        if bag_size > 0 and self.policy_network.bag_size != bag_size:
            raise ValueError("the agent's bag_size must match the network's")
        # actor staging: rolling context -> pinned -> device, Q row -> pinned
        L, O, A = context_len, int(np.prod(env_obs_length)), num_actions
        pin = (lambda t: t.pin_memory()) if cuda else (lambda t: t)
        if self.image is not None:
            # image nets act through the module forward (encoder + row-block forward): the context images are staged as uint8
            self._img_ctx_h = pin(torch.zeros(L, O, dtype=torch.uint8))
            O = 1
        # one packed staging buffer [L*O f32 | L u8] -> ONE host-to-device copy per action
        nbytes = L * O * 4 + L
        self._ctx_h = pin(torch.zeros(nbytes, dtype=torch.uint8))
        self._ctx_d = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self._ctx_obs_np = self._ctx_h.numpy()[:L * O * 4].view(np.float32).reshape(L, O)
        self._ctx_act_np = self._ctx_h.numpy()[L * O * 4:]
        self._ctx_hp, self._ctx_dp = ctypes.c_void_p(self._ctx_h.data_ptr()), ctypes.c_void_p(self._ctx_d.data_ptr())
        self._q_d = torch.zeros(L, A, device=self.device)
        self._q_p = ctypes.c_void_p(self._q_d.data_ptr())
        self._actor_ws, self._actor_ws_p = None, None
        self._q_h = pin(torch.zeros(A))
        self._q_np = self._q_h.numpy()
        self._q_hp = ctypes.c_void_p(self._q_h.data_ptr())
        # pipelined mode (begin_action / train / finish_action): the actor forward of step t+1 runs on its own
        # stream CONCURRENTLY with TD update t+1; it reads the weights produced by update t, and update t+1's
        # optimizer kernel waits for it (write-after-read on theta)
        self._theta_p = ctypes.c_void_p(self.engine.theta_pol.data_ptr())
        # all learner work is issued on the stream that is current NOW (normally the default stream); binding it spares
        # a current-stream lookup per launch in the step loop
        self._main_stream = torch.cuda.current_stream(self.device) if cuda else None
        self._main_ptr = ctypes.c_void_p(self._main_stream.cuda_stream) if cuda else None
        if cuda:
            self.engine.bind_stream(self._main_stream)
            self.replay_buffer.bind_stream(self._main_ptr, self._main_stream)
        if self.sampler == "device" and not self._separate_sample_launch:
            # latency mode: the next update's target pass inside the backward launch, policy passes as four row slices (learner.py)
            self.pipelined = self.engine.enable_pipeline(lambda rb=self.replay_buffer: rb.version)
        # the actor's stream and its two events are created by the first begin_action(): agents that never act in pipelined mode
        # (benchmarks of the update alone, evaluation) leave the process's hardware queues to the streams that are in use
        self._actor_stream, self._actor_ptr, self._ev_update_done, self._ev_actor_done = None, None, None, None
        self._actor_stream_ok = cuda
        self._actor_inflight = False
        self._actor_calls = 0
        self.pipelined = getattr(self, "pipelined", False)

    # ---- mode / context (dqn.py:102-115) -------------------------------------------------------
    @property
    def context(self) -> Context:
        return self.train_context if self.train_mode == TrainMode.TRAIN else self.eval_context

    @property
    def bag(self) -> Bag:                                               # dtqn.py:69-74
        return self.train_bag if self.train_mode == TrainMode.TRAIN else self.eval_bag

    def eval_on(self) -> None:
        self.train_mode = TrainMode.EVAL
        self.policy_network.eval()

    def eval_off(self) -> None:
        if self.train_mode != TrainMode.TRAIN or self.policy_network.training is False:
            self.policy_network.train()          # walks the whole module tree: only when the mode really changes
        self.train_mode = TrainMode.TRAIN

    # ---- actor (dtqn.py:76-160) -----------------------------------------------------------------
    def _launch_actor_forward(self, stream_ptr) -> int:
        """Stage the unpadded prefix of the rolling context and launch the batch-1 forward; returns n."""
        ctx = self.context
        n = min(ctx.max_length, ctx.timestep + 1)
        self._ctx_obs_np[:n] = ctx.obs[:n]
        self._ctx_act_np[:n] = ctx.action[:n, 0]
        eng = self.engine
        if self._actor_ws is None:       # tiled kernels' scratch, or the hand-over tiles of the two-workgroup latency mode
            need = eng.lib.dtqn_forward_workspace_floats(eng._actor_net_ref, 1)
            self._actor_ws = torch.zeros(max(1, need), dtype=torch.float32, device=self.device)
            self._actor_ws_p = ctypes.c_void_p(self._actor_ws.data_ptr()) if need > 0 else None
        # pinned context -> device, forward, Q of the LAST timestep -> pinned: one library call, all on `stream_ptr`
        # the reference's policy network is in train mode during rollouts (dqn.py:102-115): with dropout > 0 the action
        # forward drops units too; evaluation (eval_on) runs without.  Keyed by the count of actor forwards.
        self._actor_calls += 1
        rc = eng.lib.dtqn_actor_forward(eng._actor_net_ref, self._theta_p, self._ctx_hp, self._ctx_dp, n, self._q_p, self._q_hp,
                                        self._actor_ws_p, 1 if self.train_mode == TrainMode.TRAIN else 0, eng.td.dropout_seed ^ 0xAC70,
                                        self._actor_calls & 0xFFFFFFFF, stream_ptr)
        if rc == B.DEFINES["DTQN_ERR_ARG"]:
            raise AssertionError("Cannot forward, history is longer than expected.")   # dtqn.py:170-173
        if rc != 0:
            raise RuntimeError(f"dtqn_actor_forward failed with DTQN status {rc}")
        return n

    def _bag_forward(self, obs: np.ndarray, actions: np.ndarray, bag_obss: np.ndarray, bag_actions: np.ndarray) -> torch.Tensor:
        """policy_network(obs, actions, bag_obss, bag_actions) on host arrays (batch-first); Q stays on the device."""
        t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=self.device)
        drop = None
        if self.train_mode == TrainMode.TRAIN and self.policy_network.dropout_p > 0.0:
            # the reference's policy network is in train mode here (dqn.py:102-115): fresh keep masks per forward, keyed like
            # the plain actor's (seed ^ 0xAC70, count of actor forwards)
            self._actor_calls += 1
            drop = (int(self.engine.td.dropout_seed) ^ 0xAC70, self._actor_calls)
        return self.policy_network(t(obs, self.obs_tensor_type), t(actions, torch.long), t(bag_obss, self.obs_tensor_type),
                                   t(bag_actions, torch.long), _train_dropout=drop)

    @torch.no_grad()
    def _image_action(self) -> int:
        """get_action of an image net (dtqn.py:79-108): the unpadded context prefix through DTQN.forward (convolutional embedding
        + row-block forward); the policy network is in train mode during rollouts like the reference's (dropout keyed per call)."""
        ctx = self.context
        n = min(ctx.max_length, ctx.timestep + 1)
        self._img_ctx_h[:n] = torch.from_numpy(ctx.obs[:n].reshape(n, -1))
        drop = None
        if self.train_mode == TrainMode.TRAIN and self.policy_network.dropout_p > 0.0:
            self._actor_calls += 1
            drop = (int(self.engine.td.dropout_seed) ^ 0xAC70, self._actor_calls)
        obs = self._img_ctx_h[:n].to(self.device, non_blocking=True).reshape(1, n, *self.image)
        q = self.policy_network(obs, None, _train_dropout=drop)
        return int(torch.argmax(q[0, -1]).item())

    @torch.no_grad()
    def _bag_action(self) -> int:
        """get_action of a bag network (dtqn.py:79-108): the unpadded context prefix plus the WHOLE bag, padding included."""
        ctx = self.context
        n = min(ctx.max_length, ctx.timestep + 1)
        q = self._bag_forward(ctx.obs[None, :n], ctx.action[None, :n], self.bag.obss[None], self.bag.actions[None])
        return int(torch.argmax(q[:, -1, :]).item())

    @torch.no_grad()
    def get_action(self, epsilon: float = 0.0) -> int:
        if RNG.rng.random() < epsilon:
            return RNG.rng.integers(self.num_actions)
        if self.bag.size > 0:
            return self._bag_action()
        if self.image is not None:
            return self._image_action()
        self._launch_actor_forward(self.engine._stream())
        if self._main_stream is not None:
            self._main_stream.synchronize()
        return int(np.argmax(self._q_np))                                # first max, like torch.argmax

    # ---- pipelined actor (same action semantics: the policy after the previous update) ----------------
    def begin_action(self, epsilon: float = 0.0):
        """Start choosing the action for the current context without blocking.  The forward is queued on the
        actor stream behind the last TD update; call train() next (it overlaps), then finish_action()."""
        if RNG.rng.random() < epsilon:
            return int(RNG.rng.integers(self.num_actions))
        if self.bag.size > 0:                     # bag networks act through the module forward (no second stream)
            return self._bag_action()
        if self.image is not None:
            return self._image_action()
        if not self._actor_stream_ok:             # CPU kernel-emulation tests: no streams, same result
            return self._sync_forward_action()
        if self._actor_stream is None:
            self._actor_stream = torch.cuda.Stream(self.device)
            self._actor_ptr = ctypes.c_void_p(self._actor_stream.cuda_stream)
            self._ev_update_done, self._ev_actor_done = torch.cuda.Event(), torch.cuda.Event()
        if not getattr(self, "_update_recorded", True):       # order the actor behind the last TD update (pipelined mode only)
            self._ev_update_done.record(self._main_stream)
            self._update_recorded = True
        self._actor_stream.wait_event(self._ev_update_done)
        self._launch_actor_forward(self._actor_ptr)        # no torch op inside: no stream context needed
        self._ev_actor_done.record(self._actor_stream)
        self._actor_inflight = True
        return None

    def _sync_forward_action(self) -> int:
        self._launch_actor_forward(self.engine._stream())
        return int(np.argmax(self._q_np))

    def finish_action(self, pending) -> int:
        if pending is not None:
            return pending
        self._actor_stream.synchronize()
        return int(np.argmax(self._q_np))

    def context_reset(self, obs: np.ndarray) -> None:
        self.context.reset(obs)
        if self.train_mode == TrainMode.TRAIN:
            self.replay_buffer.store_obs(obs)
        if self.bag.size > 0:
            self.bag.reset()

    @torch.no_grad()
    def _bag_insert(self, bag: Bag, ctx: Context, evicted_obs, evicted_action) -> None:
        """What the context evicted goes to the bag; a full bag keeps the best of its bag_size + 1 candidate contents by the
        policy network's mean-over-time max-Q (dtqn.py:125-157).  Shared by observe() and the vectorised rollout."""
        if bag.add(evicted_obs, evicted_action):
            return
        # candidate i < bag_size: entry i replaced by the evicted pair; candidate bag_size: the bag as it is
        k = bag.size + 1
        cand_obss = np.tile(bag.obss, (k, 1, 1))
        cand_actions = np.tile(bag.actions, (k, 1, 1))
        for i in range(bag.size):
            cand_obss[i, i] = evicted_obs
            cand_actions[i, i] = evicted_action
        q = self._bag_forward(np.tile(ctx.obs, (k, 1, 1)), np.tile(ctx.action, (k, 1, 1)), cand_obss, cand_actions)
        keep = int(torch.argmax(torch.mean(torch.max(q, 2)[0], 1)).item())      # highest mean-over-time max-Q
        bag.obss, bag.actions = cand_obss[keep], cand_actions[keep]

    def observe(self, obs: np.ndarray, action: int, reward: float, done: bool) -> None:
        """Add a transition to the context; what the context evicts goes to the bag, and when the bag is full the policy
        network picks which of the bag_size + 1 candidate bags to keep (dtqn.py:116-160)."""
        evicted_obs, evicted_action = self.context.add_transition(obs, action, reward, done)
        if self.bag.size > 0 and evicted_obs is not None:
            self._bag_insert(self.bag, self.context, evicted_obs, evicted_action)
        if self.train_mode == TrainMode.TRAIN:
            self.replay_buffer.store(obs, action, reward, done, self.context.timestep)

    # ---- learner (dtqn.py:162-269) --------------------------------------------------------------
    def train(self) -> None:
        rb = self.replay_buffer
        if self.dp is None:
            if not rb.can_sample(self.batch_size):
                return
        elif not self._dp_ready:
            # every rank must enter (or skip) the gradient all-reduce together: the "nothing to sample yet" early return
            # (dtqn.py:163-164) is decided collectively until all shards can sample -- can_sample never turns false again
            if not ddp.agree_all(rb.can_sample(self.batch_size), self.device):
                return
            self._dp_ready = True
        self.eval_off()
        eng = self.engine
        sp = eng._stream()
        rb.commit_finished(sp, self._main_stream)
        if self.sampler == "reference" and self.bag.size > 0:
            eps, starts, rows = rb.sample_bag_indices(self.batch_size, self.bag.size)       # dtqn.py:166-177
            eng.set_indices(eps, starts)
            eng.gather_bag(rb.dev, rows)
        elif self.sampler == "reference":
            eng.set_indices(*rb.sample_indices(self.batch_size))
        else:
            n_valid, exclude = rb.valid_range()
            if self._separate_sample_launch:          # DTQN_SAMPLE_LAUNCH=1: A/B knob, one extra launch per update
                eng.sample_on_device(rb.dev, n_valid, exclude, self.sample_seed, sp)
                if self.bag.size > 0:             # the windows' bags: same device draw as the in-kernel sampler's
                    eng.gather_bag_on_device(rb.dev, self.sample_seed)
            else:
                eng.sample_in_forward(n_valid, exclude, self.sample_seed)      # the forward kernel draws its own windows
        if self._actor_inflight:
            # the gradient kernels overlap the actor forward; only the optimizer kernel (which overwrites theta)
            # has to wait for it
            if self.dp is not None:
                self.dp.forward_backward(rb.dev)
                self.dp.reduce()
            else:
                eng.forward_backward(rb.dev)
            self._main_stream.wait_event(self._ev_actor_done)     # (measured: dropping this wait would buy the loop 2.4 %)
            eng.clip_adam()
            self._actor_inflight = False
        elif self.dp is None:
            eng.update(rb.dev, sp)
        else:
            self.dp.update(rb.dev)
        self._update_recorded = False
        self._enqueue_stats()
        self.num_train_steps += 1
        # hard target sync happens on the device every target_update_frequency optimizer steps

    def target_update(self) -> None:
        """Hard update: target <- policy (dqn.py:208-210)."""
        self.engine.target_sync()

    # ---- asynchronous statistics ---------------------------------------------------------------
    # The optimizer kernel writes the statistics of call k to slot (k-1) % RING_SLOTS of a pinned host ring and
    # tags the slot with k last; the host only polls memory: no copy, event or sync per update.
    def _enqueue_stats(self) -> None:
        self._calls_issued += 1
        backlog = self._calls_issued - self._calls_read
        if backlog >= self.engine.RING_SLOTS - 1:
            self._drain_stats(block=True)
        elif backlog >= self.STATS_DRAIN_EVERY:     # readers (DeferredRunningAverage.mean, checkpoints) force a full drain themselves
            self._drain_stats(block=False)

    def _restart_stats_ring(self, abandon: bool = False) -> None:
        """In front of a reload of the optimizer state (optimizer.load_state_dict resets the device's call counter): finish and consume
        the outstanding updates, then restart the statistics ring -- host counters and slot tags -- with it.  abandon: the run so far is
        being replaced by a checkpoint, so an error it still had pending is not this load's business."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        try:
            self._drain_stats(block=True)
        except RuntimeError:
            if not abandon:
                raise
        self._calls_issued = self._calls_read = 0
        self.engine.stats_ring.zero_()

    def _drain_stats(self, block: bool) -> None:
        eng = self.engine
        ring, slots = eng.stats_ring_np, eng.RING_SLOTS
        i_nonfinite, idx = self._stat_index
        rows, why = [], 0
        while self._calls_read < self._calls_issued:
            k = self._calls_read + 1
            row = ring[(k - 1) % slots]               # [12][2]: {value, tag} granules, each one 8-byte store of the kernel
            tag = float(k & 0x7FFFFF)                 # the kernel tags with the call index modulo 2^23 (exact in f32)
            if not (row[:, 1] == tag).all():
                if not block:
                    break
                if self.device.type == "cuda":
                    (self._main_stream if self._main_stream is not None else torch.cuda.current_stream(self.device)).synchronize()
                if not (row[:, 1] == tag).all():
                    raise RuntimeError(f"statistics of update call {k} never arrived (ring tags {row[:, 1].tolist()})")
            vals = row[:, 0].copy()
            if not (row[:, 1] == tag).all():          # overwritten while it was copied (the ring is 256 calls deep: cannot happen
                raise RuntimeError(f"statistics ring slot of call {k} was overwritten while it was read")      # with the drain cadence)
            self._calls_read = k
            rows.append(vals)
            if vals[i_nonfinite] != 0.0:
                why = int(vals[i_nonfinite])
                break
        # RunningAverage.add per (statistic, update), in update order: same sums as one call per value, without the calls.
        # A skipped update still logs its loss and Q / target statistics -- the reference adds them before clip_grad_norm_ raises
        # (dtqn.py:245-253) -- but not its gradient norm (:263 is never reached).
        for name, i in idx:
            sink = self._stat_sinks[name]
            q, size, tot = sink.q, sink.size, sink.sum
            for vals in rows:
                if vals[i_nonfinite] != 0.0 and name == "grad_norm":
                    continue
                v = float(vals[i])
                q.append(v)
                tot += v
                if len(q) > size:
                    tot -= q.popleft()
            sink.sum = tot
        if why:
            # The optimizer kernel skipped that update and every later one (include/dtqn_hip.h, step_counter[3]): parameters, Adam
            # moments and the step count are those in front of the failing update, as after the reference's exception; the calls
            # issued since were no-ops and are taken back.
            self.num_train_steps -= self._calls_issued - (self._calls_read - 1)
            self._calls_read = self._calls_issued
            if why == 2:
                raise RuntimeError("device-side gradient exchange: a peer's gradient never arrived (bounded wait expired); "
                                   "the update was skipped on this rank")
            # torch.nn.utils.clip_grad_norm_(..., error_if_nonfinite=True) raises this in the reference (dtqn.py:257-261)
            raise RuntimeError(NONFINITE_MESSAGE)

    # ---- checkpoints (dqn.py:212-327), plain arrays instead of pickled objects -------------------
    def save_mini_checkpoint(self, checkpoint_dir: str, wandb_id: Optional[str]) -> None:
        torch.save({"step": self.num_train_steps, "wandb_id": wandb_id}, checkpoint_dir + "_mini_checkpoint.pt")

    @staticmethod
    def load_mini_checkpoint(checkpoint_dir: str) -> dict:
        return torch.load(checkpoint_dir + "_mini_checkpoint.pt")

    def save_checkpoint(self, checkpoint_dir: str, wandb_id, episode_successes: RunningAverage,
                        episode_rewards: RunningAverage, episode_lengths: RunningAverage, eps) -> None:
        self._drain_stats(block=True)
        rng_state = {"random_rng_state": random.getstate(), "rng_bit_generator_state": RNG.rng.bit_generator.state,
                     "numpy_rng_state": np.random.get_state(), "torch_rng_state": torch.get_rng_state()}
        r = ddp.rank()
        if r > 0:
            # data-parallel replica: parameters / optimizer / statistics are identical to rank 0's and are saved there;
            # this rank's OWN state is its replay shard and its RNG streams
            torch.save({"step": self.num_train_steps, "replay_buffer_pos": [self.replay_buffer.pos[0], 0], **rng_state},
                       checkpoint_dir + f"_checkpoint.rank{r}.pt")
            np.savez(checkpoint_dir + f"buffer.rank{r}.npz", **self.replay_buffer.export_arrays())
            return
        self.save_mini_checkpoint(checkpoint_dir=checkpoint_dir, wandb_id=wandb_id)
        ra = lambda r: {"size": r.size, "q": list(r.q), "sum": r.sum}
        torch.save({
            "step": self.num_train_steps, "wandb_id": wandb_id, "replay_buffer_pos": [self.replay_buffer.pos[0], 0],
            "policy_net_state_dict": self.policy_network.state_dict(), "target_net_state_dict": self.target_network.state_dict(),
            "optimizer_state_dict": self.optimizer.state_dict(), "epsilon": eps.val,
            "episode_successes": ra(episode_successes), "episode_rewards": ra(episode_rewards), "episode_lengths": ra(episode_lengths),
            **{k: ra(v) for k, v in (("td_errors", self.td_errors), ("grad_norms", self.grad_norms),
                                     ("qvalue_max", self.qvalue_max), ("qvalue_mean", self.qvalue_mean),
                                     ("qvalue_min", self.qvalue_min), ("target_max", self.target_max),
                                     ("target_mean", self.target_mean), ("target_min", self.target_min))},
            **rng_state,
        }, checkpoint_dir + "_checkpoint.pt")
        np.savez(checkpoint_dir + "buffer.npz", **self.replay_buffer.export_arrays())

    def load_checkpoint(self, checkpoint_dir: str) -> Tuple[str, RunningAverage, RunningAverage, RunningAverage, float]:
        ck = torch.load(checkpoint_dir + "_checkpoint.pt", weights_only=False)
        self._restart_stats_ring(abandon=True)
        self._actor_inflight = False
        self.num_train_steps = ck["step"]
        shard, r = ck, ddp.rank()
        buffer_file = checkpoint_dir + "buffer.npz"
        reseed = False
        if r > 0:
            # replicas resume from their own replay shard and RNG streams; without a shard (checkpoint written by a
            # smaller job) the replica takes rank 0's replay and re-seeds its streams so ranks do not draw the same batches
            if os.path.exists(checkpoint_dir + f"_checkpoint.rank{r}.pt") and os.path.exists(checkpoint_dir + f"buffer.rank{r}.npz"):
                shard = torch.load(checkpoint_dir + f"_checkpoint.rank{r}.pt", weights_only=False)
                buffer_file = checkpoint_dir + f"buffer.rank{r}.npz"
                if shard["step"] != ck["step"]:
                    raise RuntimeError(f"rank {r} checkpoint shard is from step {shard['step']}, rank 0's from {ck['step']}")
            else:
                reseed = True
        self.replay_buffer.pos = list(shard["replay_buffer_pos"])
        self.replay_buffer.import_arrays(dict(np.load(buffer_file)))
        self.policy_network.load_state_dict(ck["policy_net_state_dict"])
        self.target_network.load_state_dict(ck["target_net_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])

        def restore(dst: RunningAverage, src: dict) -> RunningAverage:
            dst.size, dst.sum = src["size"], src["sum"]
            dst.q.clear(); dst.q.extend(src["q"])
            return dst
        for k in ("td_errors", "grad_norms", "qvalue_max", "qvalue_mean", "qvalue_min", "target_max", "target_mean", "target_min"):
            restore(getattr(self, k), ck[k])
        random.setstate(shard["random_rng_state"])
        RNG.rng.bit_generator.state = shard["rng_bit_generator_state"]
        np.random.set_state(shard["numpy_rng_state"])
        torch.set_rng_state(shard["torch_rng_state"])
        if reseed:
            base = int(ck["step"]) * 1000003 + r
            random.seed(base)
            RNG.rng = np.random.Generator(np.random.PCG64(base))
            np.random.seed(base % (2 ** 32))
        self._dp_ready = False
        out = [restore(RunningAverage(10), ck[k]) for k in ("episode_successes", "episode_rewards", "episode_lengths")]
        return ck["wandb_id"], out[0], out[1], out[2], ck["epsilon"]


# north_star spells the symbol this way; the reference's own name is DtqnAgent (dtqn/agents/dtqn.py:15)
DTQNAgent = DtqnAgent
