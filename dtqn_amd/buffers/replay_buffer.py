"""Episode-major replay buffer with the reference's producer API (dtqn/buffers/replay_buffer.py),
resident in HBM.

Layout and semantics follow the reference (slot `pos[0] % max_size` is the episode in progress;
row t+1 holds the observation after step t; unwritten tail = (mask, 0, 0.0, done)), but the arrays
live on the GPU (dtqn_amd.learner.DeviceReplay).  The actor's writes are queued on the host as
small records, staged through pinned memory, copied with one async H2D per flush and scattered by
one kernel (dtqn_replay_apply), instead of five tiny host->device copies per env step.  Sampling
never gathers on the host: the TD kernels read their windows in place from (episode, start) pairs.
"""
from __future__ import annotations

import ctypes
import random
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _binding as B
from ..learner import DeviceReplay

RECORD_DTYPE = np.dtype([("kind", "<i4"), ("ep", "<i4"), ("t", "<i4"), ("action", "<i4"), ("reward", "<f4"),
                         ("done", "<i4"), ("obs_index", "<i4"), ("ep_len", "<i4")])
assert RECORD_DTYPE.itemsize == ctypes.sizeof(B.DtqnReplayRecord)


class ReplayBuffer:
    STAGE_CAPACITY = 256     # records per staging buffer
    STAGE_RING = 4

    def __init__(self, buffer_size: int, env_obs_length: int, obs_mask, max_episode_steps: int,
                 context_len: Optional[int] = 1, device=None, lib=None):
        # image observations: env_obs_length is their (C, H, W) shape; stored as uint8 rows of C*H*W bytes (replay_buffer.py:36-45)
        self.image = tuple(int(v) for v in env_obs_length) if isinstance(env_obs_length, (tuple, list)) else None
        self.max_size = buffer_size // max_episode_steps
        self.context_len = context_len
        self.env_obs_length = env_obs_length
        self.max_episode_steps = max_episode_steps
        self.obs_mask = obs_mask
        self.pos = [0, 0]
        self.device = torch.device(device)
        self._lib = lib
        self.dev = DeviceReplay(self.max_size, max_episode_steps, env_obs_length, float(obs_mask), self.device)
        # int32, not the reference's uint8 (which wraps at 256 and, under numpy >= 2, underflows in
        # `episode_lengths[idx] - context_len`)
        self.episode_lengths = np.zeros([self.max_size], dtype=np.int32)
        cuda = self.device.type == "cuda"
        if self.image is not None:
            self.STAGE_CAPACITY = 32                   # 62 KB per 144 x 144 x 3 observation: a smaller staging ring
        C, O = self.STAGE_CAPACITY, int(np.prod(env_obs_length))
        self._stage = []
        for _ in range(self.STAGE_RING if cuda else 1):
            rec_h = torch.zeros(C * RECORD_DTYPE.itemsize, dtype=torch.uint8)
            obs_h = torch.zeros(C * O, dtype=torch.uint8 if self.image is not None else torch.float32)
            if cuda:
                rec_h, obs_h = rec_h.pin_memory(), obs_h.pin_memory()
            self._stage.append(dict(
                rec_h=rec_h, obs_h=obs_h, rec_np=rec_h.numpy().view(RECORD_DTYPE), obs_np=obs_h.numpy().reshape(C, O),
                rec_d=torch.zeros_like(rec_h, device=self.device), obs_d=torch.zeros_like(obs_h, device=self.device),
                event=torch.cuda.Event() if cuda else None, busy=False, views={}))
            self._stage[-1]["rec_p"] = ctypes.c_void_p(self._stage[-1]["rec_d"].data_ptr())
            self._stage[-1]["obs_p"] = ctypes.c_void_p(self._stage[-1]["obs_d"].data_ptr())
            self._stage[-1]["rec_hp"] = ctypes.c_void_p(rec_h.data_ptr())
            self._stage[-1]["obs_hp"] = ctypes.c_void_p(obs_h.data_ptr())
        self._cur, self._n = 0, 0
        self.version = 0           # bumped whenever the device arrays change (a scatter launch, import_arrays): a window drawn ahead
                                   # of its update (TdEngine pipeline) is only used if nothing was written since

    # ---- producer side (replay_buffer.py:71-98) ----------------------------------------------
    def _push(self, kind: int, ep: int, t: int, action: int, reward: float, done: bool, obs, ep_len: int = 0) -> None:
        if self._n == self.STAGE_CAPACITY:
            self.commit()
        st = self._stage[self._cur]
        if self._n == 0 and st["busy"]:
            st["event"].synchronize()          # the copy that last used this pinned buffer has completed
            st["busy"] = False
        i = self._n
        st["obs_np"][i] = np.asarray(obs).reshape(-1) if self.image is not None else obs
        st["rec_np"][i] = (kind, ep, t, action, reward, int(bool(done)), i, ep_len)
        self._n += 1

    def store_obs(self, obs: np.ndarray) -> None:
        """First observation of an episode: cleanses the slot, then writes row 0."""
        ep = self.pos[0] % self.max_size
        self.episode_lengths[ep] = 0
        self._push(0, ep, 0, 0, 0.0, True, obs)

    def store(self, obs: np.ndarray, action, reward, done, episode_length: Optional[int] = 0) -> None:
        ep, t = self.pos[0] % self.max_size, self.pos[1]
        self.episode_lengths[ep] = episode_length
        self._push(1, ep, t, int(action), float(reward), done, obs, int(episode_length))
        self.pos = [self.pos[0], self.pos[1] + 1]

    def can_sample(self, batch_size: int) -> bool:
        return batch_size < self.pos[0]

    def bind_stream(self, stream_ptr, stream) -> None:
        """commit() calls without a stream (a full staging buffer inside store(), the samplers, export) go to this stream: the
        learner's, so that the scatter launch stays ordered with the TD updates that read the arrays."""
        self._bound_ptr, self._bound_stream = stream_ptr, stream

    def flush(self) -> None:
        self.pos = [self.pos[0] + 1, 0]
        self._episode_finished = True

    def commit_finished(self, stream_ptr=None, stream=None) -> None:
        """commit(), but only when the staged records complete an episode.  The samplers never draw from the slot in
        progress (valid_range / replay_buffer.py:141-145), so its records can wait in the pinned staging until the episode
        ends or the staging is full: one scatter launch per episode (or per STAGE_CAPACITY steps) instead of one per
        environment step in front of every TD update."""
        if getattr(self, "_episode_finished", False):
            self.commit(stream_ptr, stream)

    def commit(self, stream_ptr=None, stream=None) -> None:
        """Ship the queued records to the GPU (dtqn_replay_push) on torch's current stream, or on the raw stream
        handle `stream_ptr` when the caller already has it."""
        self._episode_finished = False
        if self._n == 0:
            return
        st, n = self._stage[self._cur], self._n
        if stream_ptr is None and getattr(self, "_bound_ptr", None) is not None:
            stream_ptr, stream = self._bound_ptr, self._bound_stream      # the learner's stream (bind_stream): same order as its updates
        if stream_ptr is None and self.device.type == "cuda":
            stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        # the scatter kernel reads the pinned staging in place: one launch, no copy
        self.version += 1
        rc = self._lib.dtqn_replay_push(self.dev.view_ref, st["rec_hp"], st["obs_hp"], n, stream_ptr)
        if rc != 0:
            raise RuntimeError(f"dtqn_replay_push failed with DTQN status {rc}")
        if st["event"] is not None:
            st["event"].record(stream) if stream is not None else st["event"].record()
            st["busy"] = True
        self._cur = (self._cur + 1) % len(self._stage)
        self._n = 0

    # ---- consumer side -----------------------------------------------------------------------
    def valid_range(self) -> Tuple[int, int]:
        """(n_valid, exclude): finished slots are [0, n_valid) minus the one in progress (:141-145)."""
        return min(self.pos[0], self.max_size), self.pos[0] % self.max_size

    def sample_indices(self, batch_size: int) -> Tuple[np.ndarray, np.ndarray]:
        """(episode, start) draws with the reference's distribution AND its use of Python's `random`
        stream (replay_buffer.py:141-158), so a seeded run visits the same windows."""
        n_valid, exclude = self.valid_range()
        valid = [i for i in range(n_valid) if i != exclude]
        eps = np.fromiter((random.choice(valid) for _ in range(batch_size)), dtype=np.int32, count=batch_size)
        L = self.context_len
        starts = np.fromiter((random.randint(0, max(0, int(self.episode_lengths[e]) - L)) for e in eps),
                             dtype=np.int32, count=batch_size)
        return eps, starts

    def sample_bag_indices(self, batch_size: int, bag_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """The draws of ReplayBuffer.sample_with_bag (replay_buffer.py:171-254) from Python's `random` stream, as indices:
        (episodes [B], starts [B], rows [B][2][bag_size]); rows[b][0] / rows[b][1] name the episode rows (all before the window)
        whose observations / actions fill bag b, -1 = unused entry (obs_mask, action 0).  Reference behaviour kept as is:
          * the episode list skips `i != pos[0]` WITHOUT the modulo of sample() (:183-187), so once the buffer has wrapped the
            slot in progress can be drawn too;
          * a window starting before row bag_size takes ALL earlier rows in order (:221-229); otherwise observation rows and
            action rows are two INDEPENDENT random.sample draws (:232-252), so a bag entry's action need not belong to its
            observation.  random.sample consumes the stream as a function of (population size, k) only, so sampling row
            numbers visits the same rows as the reference's sampling of the row contents."""
        valid = [i for i in range(min(self.pos[0], self.max_size)) if i != self.pos[0]]
        eps = np.fromiter((random.choice(valid) for _ in range(batch_size)), dtype=np.int32, count=batch_size)
        L = self.context_len
        starts = np.fromiter((random.randint(0, max(0, int(self.episode_lengths[e]) - L)) for e in eps),
                             dtype=np.int32, count=batch_size)
        rows = np.full((batch_size, 2, bag_size), -1, dtype=np.int32)
        for b in range(batch_size):
            st = int(starts[b])
            if st < bag_size:
                rows[b, :, :st] = np.arange(st)
            else:
                rows[b, 0] = random.sample(range(st), k=bag_size)
                rows[b, 1] = random.sample(range(st), k=bag_size)
        return eps, starts, rows

    def sample_with_bag(self, batch_size: int, sample_bag):
        """Reference-shaped sample_with_bag() -> numpy arrays (compatibility / debugging path, like sample(); the TD update
        gathers windows and bags on the device from sample_bag_indices)."""
        self.commit()
        eps, starts, rows = self.sample_bag_indices(batch_size, sample_bag.size)
        e = torch.as_tensor(eps, dtype=torch.long, device=self.device).unsqueeze(1)
        tr = torch.as_tensor(starts, dtype=torch.long, device=self.device).unsqueeze(1) + \
            torch.arange(self.context_len, device=self.device).unsqueeze(0)
        d = self.dev
        out = (d.obs[e, tr], d.actions[e, tr].unsqueeze(-1), d.rewards[e, tr].unsqueeze(-1), d.obs[e, tr + 1],
               d.actions[e, tr + 1].unsqueeze(-1), d.dones[e, tr].unsqueeze(-1).bool())
        r = torch.as_tensor(rows, dtype=torch.long, device=self.device)
        bo = torch.where((r[:, 0] >= 0).unsqueeze(-1), d.obs[e, r[:, 0].clamp(min=0)], torch.full((), float(self.obs_mask), device=self.device))
        ba = torch.where(r[:, 1] >= 0, d.actions[e, r[:, 1].clamp(min=0)], torch.zeros((), dtype=d.actions.dtype, device=self.device))
        return tuple(x.cpu().numpy() for x in out) + (self.episode_lengths[eps.reshape(-1, 1)], bo.cpu().numpy(),
                                                      ba.unsqueeze(-1).cpu().numpy().astype(np.int64))

    def sample(self, batch_size: int):
        """Reference-shaped sample() -> numpy arrays (compatibility / debugging path: it gathers on
        the device and copies back; the TD update does not use it)."""
        self.commit()
        eps, starts = self.sample_indices(batch_size)
        e = torch.as_tensor(eps, dtype=torch.long, device=self.device).unsqueeze(1)
        tr = torch.as_tensor(starts, dtype=torch.long, device=self.device).unsqueeze(1) + \
            torch.arange(self.context_len, device=self.device).unsqueeze(0)
        d = self.dev
        shp = (lambda x: x.reshape(*x.shape[:2], *self.image)) if self.image is not None else (lambda x: x)
        out = (shp(d.obs[e, tr]), d.actions[e, tr].unsqueeze(-1), d.rewards[e, tr].unsqueeze(-1), shp(d.obs[e, tr + 1]),
               d.actions[e, tr + 1].unsqueeze(-1), d.dones[e, tr].unsqueeze(-1).bool())
        lens = np.clip(self.episode_lengths[eps.reshape(-1, 1)], 0, self.context_len)
        return tuple(x.cpu().numpy() for x in out) + (lens,)

    # ---- whole-array access (checkpointing) ---------------------------------------------------
    def export_arrays(self) -> dict:
        self.commit()
        d = self.dev
        return dict(obss=d.obs.cpu().numpy(), actions=d.actions.cpu().numpy(), rewards=d.rewards.cpu().numpy(),
                    dones=d.dones.cpu().numpy(), eplens=self.episode_lengths.copy())

    def import_arrays(self, arrays: dict) -> None:
        """Overwrite the device arrays (checkpoint load, synthetic fills).  Records still queued in the host staging
        belong to the state being replaced and are dropped; shapes must match this buffer's (E, T+1, O) geometry."""
        d = self.dev
        E, T, O = self.max_size, self.max_episode_steps, int(np.prod(self.env_obs_length))
        if self.image is not None:
            arrays = dict(arrays, obss=np.asarray(arrays["obss"]).reshape(E, T + 1, O))     # (E, T+1, C, H, W) of the reference's buffer is fine too
        want = {"obss": (E, T + 1, O), "actions": (E, T + 1), "rewards": (E, T), "dones": (E, T), "eplens": (E,)}
        for k, shape in want.items():
            if tuple(np.shape(arrays[k])) != shape:
                raise ValueError(f"replay array {k!r} has shape {tuple(np.shape(arrays[k]))}, this buffer needs {shape}")
        self._n = 0
        self.version += 1
        as_t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt))
        d.obs.copy_(as_t(arrays["obss"], np.uint8 if self.image is not None else np.float32)); d.actions.copy_(as_t(arrays["actions"], np.uint8))
        d.rewards.copy_(as_t(arrays["rewards"], np.float32)); d.dones.copy_(as_t(arrays["dones"], np.uint8))
        self.episode_lengths[:] = arrays["eplens"]
        d.ep_len.copy_(torch.from_numpy(self.episode_lengths))
