"""Device-side state of one DTQN learner and the launch sequence of a TD update.

`TdEngine` owns the flat parameter / optimizer buffers, the replay views and the workspaces the
kernels need, as torch tensors (torch is the allocator and stream provider here, nothing more),
and drives libdtqn_hip.so through the C ABI of include/dtqn_hip.h.  It is the body of
DtqnAgent.train() (dtqn/agents/dtqn.py:162-269 in the reference) minus the host bookkeeping.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _binding as B
from . import engine

STAT_NAMES = ("td_error", "grad_norm", "qvalue_max", "qvalue_mean", "qvalue_min",
              "target_max", "target_mean", "target_min", "clip_coef", "step", "target_synced", "nonfinite")
OPT_BLOCK_ELEMS = 1024        # dtqn_optim.hip: 256 threads x float4


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class DeviceReplay:
    """Episode-major replay arrays resident in HBM (layout: include/dtqn_hip.h, DtqnReplay)."""

    def __init__(self, num_episodes: int, max_steps: int, obs_dim, obs_mask: float, device):
        # obs_dim: an int, or the (C, H, W) shape of image observations -- stored as uint8 like the reference (replay_buffer.py:36-45)
        self.image = tuple(int(v) for v in obs_dim) if isinstance(obs_dim, (tuple, list)) else None
        E, T, O = int(num_episodes), int(max_steps), int(np.prod(obs_dim))
        self.E, self.T, self.O, self.obs_mask = E, T, O, float(obs_mask)
        self.device = device
        # initial fill = the reference constructor's (replay_buffer.py:36-69)
        self.obs = torch.full((E, T + 1, O), int(obs_mask) if self.image else float(obs_mask),
                              dtype=torch.uint8 if self.image else torch.float32, device=device)
        self.actions = torch.zeros((E, T + 1), dtype=torch.uint8, device=device)
        self.rewards = torch.zeros((E, T), dtype=torch.float32, device=device)
        self.dones = torch.ones((E, T), dtype=torch.uint8, device=device)
        self.ep_len = torch.zeros((E,), dtype=torch.int32, device=device)
        self.view = B.DtqnReplay()
        self.view.obs, self.view.actions = (None if self.image else self.obs.data_ptr()), self.actions.data_ptr()
        self.view.obs_u8 = self.obs.data_ptr() if self.image else None
        self.view.rewards, self.view.dones = self.rewards.data_ptr(), self.dones.data_ptr()
        self.view.ep_len = self.ep_len.data_ptr()
        self.view.num_episodes, self.view.max_steps, self.view.obs_dim = E, T, O
        self.view.obs_mask = float(obs_mask)
        self.view_ref = ctypes.byref(self.view)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.obs, self.actions, self.rewards, self.dones, self.ep_len))


class TdEngine:
    def __init__(self, net: B.DtqnNet, batch: int, *, lr=3e-4, gamma=0.99, history=None, tuf=10_000,
                 grad_norm_clip=1.0, betas=(0.9, 0.999), eps=1e-8, n_split: Optional[int] = None, dropout_seed: int = 0,
                 device=None, _test_lib=None, theta_pol: Optional[torch.Tensor] = None,
                 theta_tgt: Optional[torch.Tensor] = None):
        # `_test_lib` exists for the CPU kernel-emulation tests only (tests/emu); the product path
        # always resolves to the hipcc-built engine on a ROCm device and raises otherwise.
        if _test_lib is None:
            self.lib = engine.get_lib()
            self.device = engine.require_gpu() if device is None else torch.device(device)
            if self.device.type != "cuda":
                raise engine.EngineUnavailable("TdEngine needs a ROCm device; there is no CPU path")
        else:
            self.lib = _test_lib
            self.device = torch.device("cpu")
        # `actor_net`: the net of the caller's modules (what acting / inference forwards use).  `net`: what the TD update runs on --
        # the same, or its row-block twin where the library's policy says that family is faster for this batch (same theta
        # layout, records laid out for the tiled kernels; include/dtqn_hip.h, dtqn_td_prefers_tiled)
        self.actor_net = net
        self.batch = int(batch)
        if self.lib.dtqn_td_prefers_tiled(ctypes.byref(net), self.batch):
            twin = B.DtqnNet()
            if (self.lib.dtqn_net_tiled_twin(ctypes.byref(net), ctypes.byref(twin)) == 0 and twin.n_theta == net.n_theta
                    and twin.n_trainable == net.n_trainable and twin.off_pos == net.off_pos and B.param_table(twin) == B.param_table(net)):
                net = twin
            elif not net.tiled and B.ws_lite(net):
                # head width 32 / width-padded shapes have whole-sequence kernels for latency-mode batches only: without the twin the
                # first update would end in DTQN_ERR_CONFIG with no reason given
                raise ValueError(f"batch {self.batch} of this network shape trains on the row-block kernels only and its row-block twin "
                                 "could not be built (dtqn_net_tiled_twin): use a batch inside latency mode or a shape the twin covers")
        self.net = net
        self._bound_stream = None
        dev = self.device
        nt, nth = net.n_trainable, net.n_theta
        f32 = dict(dtype=torch.float32, device=dev)
        if theta_pol is None:
            frozen = np.zeros(nth, dtype=np.float32)
            self.lib.dtqn_net_fill_frozen(ctypes.byref(net), frozen.ctypes.data_as(ctypes.c_void_p))
            self.theta_pol = torch.from_numpy(frozen).to(dev)
            self.theta_tgt = torch.from_numpy(frozen.copy()).to(dev)
        else:
            # bind to the flat buffers of the caller's DTQN modules (dtqn_amd.networks.dtqn.DTQN.flat)
            same_dev = lambda a, b: a.type == b.type and (a.type != "cuda" or (a.index if a.index is not None else torch.cuda.current_device()) == (b.index if b.index is not None else torch.cuda.current_device()))
            for th in (theta_pol, theta_tgt):
                if not same_dev(th.device, dev) or th.dtype != torch.float32 or th.numel() != nth or not th.is_contiguous():
                    raise ValueError("theta buffers must be contiguous float32 tensors of n_theta elements on the engine device")
            self.theta_pol, self.theta_tgt = theta_pol, theta_tgt
        self.grad = torch.zeros(nt, **f32)
        self.adam_m = torch.zeros(nt, **f32)
        self.adam_v = torch.zeros(nt, **f32)
        Bn = self.batch
        if n_split is None:   # enough workgroups to fill the chip once (include/dtqn_hip.h: min(batch, 16), or one round of the 128 x 128 tiles)
            n_split = max(1, int(self.lib.dtqn_td_wgrad_splits(ctypes.byref(net), Bn)))
        self.n_split = int(n_split)
        self.n_norm_blocks = (nt + OPT_BLOCK_ELEMS - 1) // OPT_BLOCK_ELEMS
        # the tiled path keeps the records of all three forwards (policy(o), policy(o'), target(o')); the
        # whole-sequence kernels only save the training forward's
        self.act = torch.zeros((3 if net.tiled else 1) * Bn * net.act_stride, **f32)
        self.grd = torch.zeros(Bn * net.grd_stride, **f32)
        # latency mode for small batches: two workgroups per sequence (include/dtqn_hip.h, dtqn_td_row_split)
        self.row_split = int(self.lib.dtqn_td_row_split(ctypes.byref(net), Bn))
        RS = self.row_split
        self.small = torch.zeros(Bn * RS * net.sp_parts * net.sp_stride, **f32)
        self.q3 = torch.zeros(3 * Bn * net.lp * net.ap, **f32)
        self.gsplit = torch.zeros(self.n_split * nt, **f32)
        self.norm_partial = torch.zeros(max(self.n_norm_blocks, int(self.lib.dtqn_td_norm_partials(ctypes.byref(net)))), **f32)
        self.stats_partial = torch.zeros(Bn * RS * 8, **f32)
        self.xch = torch.zeros(max(1, self.lib.dtqn_td_xch_floats(ctypes.byref(net), Bn)), **f32)
        self.xflags = torch.zeros(max(1, self.lib.dtqn_td_xch_flags(ctypes.byref(net), Bn)), dtype=torch.int32, device=dev)
        self.stats = torch.zeros(len(STAT_NAMES), **f32)
        self.step_counter = torch.zeros(4, dtype=torch.int32, device=dev)
        self.RING_SLOTS = 256
        ring = torch.zeros(self.RING_SLOTS, len(STAT_NAMES), 2, dtype=torch.float32)      # {value, tag} granules (include/dtqn_hip.h)
        self.stats_ring = ring.pin_memory() if dev.type == "cuda" else ring      # written by the optimizer kernel, polled by the host
        self.stats_ring_np = self.stats_ring.numpy()
        # (episode, start) pairs: one [2][B] device array, so host-drawn pairs travel in ONE copy out of a small pinned ring
        self._idx_dev = torch.zeros(2, Bn, dtype=torch.int32, device=dev)
        self.ep_idx, self.start = self._idx_dev[0], self._idx_dev[1]
        self.IDX_RING = 8
        self._idx_ring, self._idx_i = [], 0
        for _ in range(self.IDX_RING if dev.type == "cuda" else 1):
            h = torch.zeros(2, Bn, dtype=torch.int32)
            if dev.type == "cuda":
                h = h.pin_memory()
            self._idx_ring.append(dict(h=h, np=h.numpy(), event=torch.cuda.Event() if dev.type == "cuda" else None, busy=False))
        self._bound_torch_stream = None
        jobs = (B.DtqnWJob * net.n_wjobs)()
        rc = self.lib.dtqn_net_wjobs(ctypes.byref(net), jobs)
        if rc != 0:
            raise RuntimeError(f"dtqn_net_wjobs rc={rc}")
        jb = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
        self.wjobs = torch.from_numpy(jb).to(dev)
        td = B.DtqnTd()
        td.theta_pol, td.theta_tgt = self.theta_pol.data_ptr(), self.theta_tgt.data_ptr()
        td.grad, td.adam_m, td.adam_v = self.grad.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()
        td.ep_idx, td.start = self.ep_idx.data_ptr(), self.start.data_ptr()
        td.act, td.grd, td.small, td.q3 = self.act.data_ptr(), self.grd.data_ptr(), self.small.data_ptr(), self.q3.data_ptr()
        td.gsplit, td.norm_partial = self.gsplit.data_ptr(), self.norm_partial.data_ptr()
        td.stats_partial, td.stats = self.stats_partial.data_ptr(), self.stats.data_ptr()
        td.stats_ring, td.stats_ring_slots = self.stats_ring.data_ptr(), self.RING_SLOTS
        td.step_counter, td.wjobs = self.step_counter.data_ptr(), self.wjobs.data_ptr()
        td.xch, td.xflags, td.row_split = self.xch.data_ptr(), self.xflags.data_ptr(), RS
        td.batch = Bn
        td.history = int(net.ctx_len if history is None else history)
        td.n_split, td.n_norm_blocks = self.n_split, self.n_norm_blocks
        td.target_update_frequency = int(tuf)
        td.gamma, td.lr, td.beta1, td.beta2, td.eps = float(gamma), float(lr), float(betas[0]), float(betas[1]), float(eps)
        td.grad_norm_clip, td.grad_scale = float(grad_norm_clip), 1.0
        if net.bag_size > 0:     # the sampled windows' bags (ReplayBuffer.sample_with_bag): one bag per sequence, shared by the three forwards
            self.bag_obs = torch.zeros(Bn, net.bag_size, net.obs_dim, **f32)
            self.bag_actions = torch.zeros(Bn, net.bag_size, dtype=torch.uint8, device=dev)
            td.bag_obs, td.bag_actions = self.bag_obs.data_ptr(), self.bag_actions.data_ptr()
            self._bag_rows_dev = torch.zeros(Bn, 2, net.bag_size, dtype=torch.int32, device=dev)
            # pinned slots for host-drawn bag rows: allocated up front, walked by their own counter
            self._bag_rows_ring = [dict(h=torch.zeros(Bn, 2, net.bag_size, dtype=torch.int32).pin_memory(), event=torch.cuda.Event(), busy=False)
                                   for _ in range(self.IDX_RING)] if dev.type == "cuda" else []
            self._bag_rows_i = 0
        # row-block networks: fragment-major copies of the layer matrices (include/dtqn_hip.h, dtqn_td_wpack), rewritten by every update's
        # forward from theta_pol / theta_tgt; 0 floats where the plan does not cover the network
        n_pack = int(self.lib.dtqn_td_wpack_floats(ctypes.byref(net)))
        if n_pack > 0:
            self.wpack_pol = torch.zeros(n_pack, **f32)
            self.wpack_tgt = torch.zeros(n_pack, **f32)
            td.wpack_pol, td.wpack_tgt = self.wpack_pol.data_ptr(), self.wpack_tgt.data_ptr()
        td.dropout_seed = int(dropout_seed) & 0xFFFFFFFF      # keep masks: hash of (seed, optimizer step, pass, sequence, site, element)
        self.img = None
        if net.img_c > 0:
            # image observations: the convolutional embedding runs in front of / behind the row-block update (dtqn_amd/image.py)
            from .image import ImageEncoder
            L1 = net.ctx_len + 1
            self.img = ImageEncoder(self.lib, net, dev)
            self.img_tgt = ImageEncoder(self.lib, net, dev)           # the target network's own transposed weights
            self.xemb = torch.zeros(3 * Bn * net.lp * net.d_model, **f32)
            td.xemb = self.xemb.data_ptr()
            i32 = dict(dtype=torch.int32, device=dev)
            self._img_lists = [torch.zeros(Bn * L1, **i32) for _ in range(6)]   # pol index / dst0 / dst1 / dsrc, tgt index / dst0
            self._img_tgt_version = -1
        self.td = td
        self._net_ref, self._td_ref = ctypes.byref(self.net), ctypes.byref(td)
        # latency mode: the backward launch also computes the weight gradients (dtqn_td_wgrad then launches nothing)
        self.wgrad_fused = bool(self.lib.dtqn_td_wgrad_is_fused(self._net_ref, self._td_ref))
        self._actor_net_ref = ctypes.byref(self.actor_net)

    # -- helpers ------------------------------------------------------------------------------
    def _stream(self):
        """HIP stream the kernels are issued on: torch's current stream, or the one pinned with bind_stream()."""
        if self._bound_stream is not None:
            return self._bound_stream
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def bind_stream(self, stream: "Optional[torch.cuda.Stream]") -> None:
        """Issue every launch of this engine on `stream` from now on (None: back to torch's current stream).  Saves the
        per-call current-stream lookup in tight loops; the caller then keeps its torch work on the same stream."""
        self._bound_stream = None if stream is None else ctypes.c_void_p(stream.cuda_stream)
        self._bound_torch_stream = stream

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed with DTQN status {rc}")

    def workspace_bytes(self) -> int:
        ts = (self.act, self.grd, self.small, self.q3, self.gsplit, self.norm_partial, self.stats_partial, self.xch)
        return sum(t.numel() * t.element_size() for t in ts)

    def set_indices(self, ep_idx, start):
        """Host-drawn (episode, start) pairs (reference RNG stream) -> device."""
        self.td.sample_in_kernel = 0
        slot = self._idx_ring[self._idx_i % len(self._idx_ring)]
        self._idx_i += 1
        if slot["busy"]:
            slot["event"].synchronize()              # the copy that last read this pinned slot has completed
            slot["busy"] = False
        slot["np"][0] = ep_idx
        slot["np"][1] = start
        if self.device.type != "cuda":
            self._idx_dev.copy_(slot["h"])
            return
        # pinned -> device, asynchronous, on the stream the kernels are launched on (ordered before dtqn_td_forward)
        bound = self._bound_torch_stream
        if bound is None or bound == torch.cuda.current_stream(self.device):
            self._idx_dev.copy_(slot["h"], non_blocking=True)
            slot["event"].record()
        else:
            with torch.cuda.stream(bound):
                self._idx_dev.copy_(slot["h"], non_blocking=True)
                slot["event"].record()
        slot["busy"] = True

    def gather_bag(self, replay: "DeviceReplay", rows) -> None:
        """Fill the bags of the windows set_indices named from host-drawn episode rows [B][2][bag_size]
        (ReplayBuffer.sample_bag_indices); queued on the engine's stream behind the index copy."""
        if self.net.bag_size <= 0:
            raise RuntimeError("this network has no bag")
        h = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int32).reshape(self._bag_rows_dev.shape))
        if self.device.type == "cuda":
            # small pinned ring: the async copy must not read a buffer the next draw overwrites
            slot = self._bag_rows_ring[self._bag_rows_i % len(self._bag_rows_ring)]
            self._bag_rows_i += 1
            if slot["busy"]:
                slot["event"].synchronize()
            slot["h"].copy_(h)
            bound = self._bound_torch_stream
            with torch.cuda.stream(bound if bound is not None else torch.cuda.current_stream(self.device)):
                self._bag_rows_dev.copy_(slot["h"], non_blocking=True)
                slot["event"].record()
            slot["busy"] = True
        else:
            self._bag_rows_dev.copy_(h)
        self._check(self.lib.dtqn_replay_gather_bag(replay.view_ref, _p(self.ep_idx), _p(self.start), _p(self._bag_rows_dev), self.batch,
                                                    self.net.bag_size, 0, None, _p(self.bag_obs), _p(self.bag_actions), self._stream()),
                    "dtqn_replay_gather_bag")

    def gather_bag_on_device(self, replay: "DeviceReplay", seed: int) -> None:
        """Bags of windows that dtqn_replay_sample drew in its own launch (sample_on_device): the rows are drawn on the device
        too, keyed by (seed, step counter) exactly as dtqn_td_forward draws them when it samples in-kernel."""
        if self.net.bag_size <= 0:
            raise RuntimeError("this network has no bag")
        self._check(self.lib.dtqn_replay_gather_bag(replay.view_ref, _p(self.ep_idx), _p(self.start), None, self.batch, self.net.bag_size,
                                                    ctypes.c_uint32(seed & 0xFFFFFFFF), _p(self.step_counter), _p(self.bag_obs),
                                                    _p(self.bag_actions), self._stream()), "dtqn_replay_gather_bag")

    def set_bag(self, bag_obs, bag_actions) -> None:
        """Bags of the windows set_indices named: bag_obs [B, bag_size, obs_dim], bag_actions [B, bag_size(, 1)]."""
        if self.net.bag_size <= 0:
            raise RuntimeError("this network has no bag")
        self.bag_obs.copy_(torch.as_tensor(bag_obs, dtype=torch.float32).reshape(self.bag_obs.shape), non_blocking=True)
        self.bag_actions.copy_(torch.as_tensor(bag_actions).reshape(self.bag_actions.shape).to(torch.uint8), non_blocking=True)

    def sample_in_forward(self, n_valid: int, exclude: int, seed: int) -> None:
        """Let the next dtqn_td_forward / dtqn_td_update draw its own windows (no sampling launch): the same draw as
        sample_on_device for the same (seed, step counter)."""
        td = self.td
        td.sample_in_kernel, td.sample_n_valid, td.sample_exclude, td.sample_seed = 1, int(n_valid), int(exclude), seed & 0xFFFFFFFF

    def sample_on_device(self, replay: DeviceReplay, n_valid: int, exclude: int, seed: int, stream=None):
        self.td.sample_in_kernel = 0
        self._check(self.lib.dtqn_replay_sample(ctypes.byref(replay.view), int(n_valid), int(exclude), self.net.ctx_len,
                                                self.batch, ctypes.c_uint32(seed & 0xFFFFFFFF), _p(self.step_counter),
                                                _p(self.ep_idx), _p(self.start), stream if stream is not None else self._stream()),
                    "dtqn_replay_sample")

    # -- pipelined update (latency mode; include/dtqn_hip.h: dtqn_td_forward_part, dtqn_td_backward_ahead) -------------------------
    # At batch 32 the forward is bound by the length of one workgroup's stage chain, and 3 B 2 = 192 workgroups leave a quarter of
    # the chip idle.  The target pass of update k + 1 depends on nothing update k computes (theta_tgt only moves at a hard sync, the
    # window draw is a pure function of (seed, step, episode lengths)), so update k's BACKWARD LAUNCH carries it as 4 B extra
    # workgroups next to its own 4 B (the chain leaves half the compute units idle), and the forward of update k + 1 is the two
    # policy passes as 2 B 4 = 256 workgroups of 16 rows -- every compute unit busy, half the rows per workgroup: 41.8 -> 30.1 us
    # per launch (rocprofv3), the backward launch unchanged.  (First built with the pass on a second stream: the two event operations
    # that needs on the update's stream cost 11.6 us per update -- more than the gain.)  A pass computed ahead is only USED if what it
    # read is still what the update would read: same (n_valid, exclude, seed), same draw step, no replay write and no target sync
    # since; otherwise the target pass runs inline (same kernel, same slices: the numbers do not depend on which way it went).
    def enable_pipeline(self, replay_version) -> bool:
        """Switch the update to {policy passes as four row slices, next update's target pass inside the backward launch}.  Returns
        False (and changes nothing) where the shape is not covered.  `replay_version` (required): callable that changes whenever the
        replay arrays are written (ReplayBuffer.version) -- a pass computed ahead on an older replay must not be used."""
        if not callable(replay_version):
            raise TypeError("enable_pipeline needs a callable that changes whenever the replay arrays are written")
        import os
        if self.net.bag_size > 0 or self.img is not None or self.net.dropout > 0 or os.environ.get("DTQN_PIPELINE", "1") == "0":
            return False
        # (row_split 4 past latency mode -- 43 ... 64 sequences -- slices the backward only: measured 6.8 k updates/s pipelined against 7.9 k
        #  with whole-sequence forward passes at batch 64; dtqn_td_latency_mode)
        ride = (not self.net.tiled and self.row_split == 4 and not self.wgrad_fused and bool(self.lib.dtqn_td_fwd_slices4_ok(self._net_ref))
                and bool(self.lib.dtqn_td_latency_mode(self._net_ref, self.batch)))
        # Row-block tiled nets at small batches (BASELINE config 5: 32 sequences x 4 row blocks = 128 workgroups per pass and per
        # launch of the backward -- half the chip): the same idea on a SECOND STREAM.  The target pass of update k + 1 runs its
        # kernels beside update k's backward kernels; update k + 1's forward launches cover two passes (one round of 256 workgroups
        # instead of one and a half).  The two event operations per update that cost 11.6 us are 0.5 % of a 2 ms update here.
        side_max = int(os.environ.get("DTQN_SIDE_STREAM_MAX", "128"))       # 64-row blocks per pass up to which the side stream is used
        stream_flavour = (self.net.tiled == 1 and self.device.type == "cuda" and self.batch * (self.net.lp // 64) <= side_max)
        if not ride and not stream_flavour:
            return False
        self._qbuf = [self.q3, torch.zeros_like(self.q3)]
        nxt = B.DtqnTd()
        ctypes.memmove(ctypes.byref(nxt), ctypes.byref(self.td), ctypes.sizeof(B.DtqnTd))
        # the pass ahead runs WHILE the chain of the current update uses td.xch / td.xflags: it gets its own
        self._next_xch = torch.zeros_like(self.xch)
        self._next_xflags = torch.zeros_like(self.xflags)
        nxt.xch, nxt.xflags = self._next_xch.data_ptr(), self._next_xflags.data_ptr()
        if not ride:       # its own window indices (the multi-kernel path keeps them in memory) and stream / events
            self.td.side_stream = 1                        # (the backward leaves the other stream's kernels room: include/dtqn_hip.h)
            self._next_idx = torch.zeros_like(self._idx_dev)
            nxt.ep_idx, nxt.start = self._next_idx[0].data_ptr(), self._next_idx[1].data_ptr()
        self._pipe = dict(nxt=nxt, nxt_ref=ctypes.byref(nxt), ahead=None, steps=int(self.step_counter[1].item()), tgt_version=0,
                          replay_version=replay_version, ride=ride,
                          stream=None if ride else torch.cuda.Stream(self.device), ev_fwd=None if ride else torch.cuda.Event(),
                          ev_side=None if ride else torch.cuda.Event(), side_busy=False,
                          launch_ahead=os.environ.get("DTQN_PIPELINE", "1") != "inline",     # inline: same kernels, target pass never ahead (A/B, tests)
                          used=0, inline=0)
        return True

    def pipeline_reset(self) -> None:
        """The optimizer state was replaced (checkpoint load) or stepped behind this object's back: re-read the step count, drop
        what was computed ahead."""
        if getattr(self, "_pipe", None) is not None:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self._pipe.update(ahead=None, steps=int(self.step_counter[1].item()), side_busy=False)

    def _pipe_begin(self, replay: DeviceReplay):
        """(draw step d, have_target, td_next reference or None) of the update about to be launched; books the pass it launches."""
        pipe, td = self._pipe, self.td
        d = pipe["steps"]                                  # step_counter[1] when this update's kernels run: the key of its draw
        # theta_tgt._version: host-side writes to the target parameters that do not come through target_sync (load_state_dict on the
        # target module, an in-place torch op on one of its views) bump torch's version counter of the flat buffer
        key = (d, td.sample_n_valid, td.sample_exclude, td.sample_seed, pipe["replay_version"](), pipe["tgt_version"], replay.view.obs,
               self.theta_tgt._version)
        td.q3 = self._qbuf[d & 1].data_ptr()
        self.q3 = self._qbuf[d & 1]
        have = pipe["ahead"] == key
        pipe["used" if have else "inline"] += 1
        pipe["ahead"] = None
        # ahead for the next update: not across a hard target sync (the optimizer launch of THIS update writes theta_tgt then)
        tuf = td.target_update_frequency
        if tuf > 0 and (d + 1) % tuf == 0:
            pipe["tgt_version"] += 1
            return d, have, None
        if not pipe["launch_ahead"]:
            return d, have, None
        nxt = pipe["nxt"]
        nxt.sample_in_kernel, nxt.sample_n_valid, nxt.sample_exclude, nxt.sample_seed = 1, td.sample_n_valid, td.sample_exclude, td.sample_seed
        nxt.q3 = self._qbuf[(d + 1) & 1].data_ptr()        # its last reader, the backward of update d - 1, is behind us on the stream
        pipe["ahead"] = (d + 1,) + key[1:]
        return d, have, pipe["nxt_ref"]

    def _forward_stage(self, replay: DeviceReplay, s) -> None:
        """The three forwards of this update (staged callers: data parallel, overlapped actor)."""
        if getattr(self, "_pipe", None) is None or not self.td.sample_in_kernel:
            self._pipe_next = None
            self._check(self.lib.dtqn_td_forward(self._net_ref, replay.view_ref, self._td_ref, s), "dtqn_td_forward")
            return
        lib, n, r, t = self.lib, self._net_ref, replay.view_ref, self._td_ref
        d, have, nxt = self._pipe_begin(replay)
        if not self._pipe["ride"]:
            self._forward_stage_streams(replay, s, d, have, nxt)
            return
        # every pass of an update is keyed by the same explicit step: the three draws agree whatever the device counter holds
        self._check(lib.dtqn_td_forward_part(n, r, t, 0, 2, 4, d, s), "dtqn_td_forward_part")
        if not have:
            self._check(lib.dtqn_td_forward_part(n, r, t, 2, 1, 4, d, s), "dtqn_td_forward_part")
        self._pipe_next = (nxt, d + 1)

    def _forward_stage_streams(self, replay: DeviceReplay, s, d, have, nxt) -> None:
        """Second-stream flavour (row-block nets): policy passes on the update's stream, the NEXT update's target pass on the side
        stream, gated behind them (its kernels then run beside this update's backward kernels)."""
        lib, n, r, t, pipe = self.lib, self._net_ref, replay.view_ref, self._td_ref, self._pipe
        main = self._main_torch_stream()
        self._check(lib.dtqn_td_forward_part(n, r, t, 0, 2, 1, d, s), "dtqn_td_forward_part")
        if pipe["side_busy"]:
            main.wait_event(pipe["ev_side"])               # the pass launched ahead has left the buffers, used or not; it ran beside the
            pipe["side_busy"] = False                      # previous backward, so the wait is over before the stream gets here
        if not have:
            self._check(lib.dtqn_td_forward_part(n, r, t, 2, 1, 1, d, s), "dtqn_td_forward_part")
        self._pipe_next = None                             # the backward launch of this flavour carries nothing
        if nxt is None:
            return
        st = pipe["stream"]
        pipe["ev_fwd"].record(main)
        st.wait_event(pipe["ev_fwd"])
        self._check(lib.dtqn_td_forward_part(n, r, nxt, 2, 1, 1, d + 1, ctypes.c_void_p(st.cuda_stream)), "dtqn_td_forward_part")
        pipe["ev_side"].record(st)
        pipe["side_busy"] = True

    def _main_torch_stream(self):
        return self._bound_torch_stream if self._bound_torch_stream is not None else torch.cuda.current_stream(self.device)

    def _backward_stage(self, replay: DeviceReplay, s) -> None:
        nxt = getattr(self, "_pipe_next", None)
        if nxt is not None and nxt[0] is not None:
            self._check(self.lib.dtqn_td_backward_ahead(self._net_ref, replay.view_ref, self._td_ref, nxt[0], nxt[1], s), "dtqn_td_backward_ahead")
        else:
            self._check(self.lib.dtqn_td_backward(self._net_ref, replay.view_ref, self._td_ref, s), "dtqn_td_backward")

    # -- the update, whole or in stages (stages are what the data-parallel wrapper interleaves) --
    def update(self, replay: DeviceReplay, stream=None):
        if self.img is not None:          # image nets: the staged sequence (the one-call entry point has no encoder in it)
            self.forward_backward(replay)
            self.clip_adam()
            return
        s = stream if stream is not None else self._stream()
        if getattr(self, "_pipe", None) is not None and self.td.sample_in_kernel:
            if not self._pipe["ride"]:
                self.forward_backward(replay)
                self.clip_adam()
                return
            d, have, nxt = self._pipe_begin(replay)
            self._check(self.lib.dtqn_td_update_pipelined(self._net_ref, replay.view_ref, self._td_ref, nxt, 1 if have else 0, d, s),
                        "dtqn_td_update_pipelined")
            self._pipe["steps"] += 1
            return
        self._check(self.lib.dtqn_td_update(self._net_ref, replay.view_ref, self._td_ref, s), "dtqn_td_update")
        if getattr(self, "_pipe", None) is not None:
            # the one-call update stepped the device's optimizer counter too (host-drawn windows on an engine whose pipeline is
            # enabled): the mirror follows, so the next in-kernel draw and the predicted hard target sync stay keyed by the device's
            # step; whatever was computed ahead was keyed by the step this update has just used up
            self._pipe["steps"] += 1
            self._pipe["ahead"] = None

    def _img_encode_windows(self, replay: DeviceReplay, s):
        """Image nets, in front of dtqn_td_forward: token lists of the sampled windows, then the convolutional embedding of the
        policy rows 0..L (each row serves policy(o) and policy(o')) and of the target rows 1..L -> td.xemb."""
        net, Bn, L = self.net, self.batch, self.net.ctx_len
        if self.td.sample_in_kernel:
            # the windows must exist before the encoder runs: draw them with the same counter-based draw in their own launch
            self.sample_on_device(replay, self.td.sample_n_valid, self.td.sample_exclude, self.td.sample_seed, s)
        pi, pd0, pd1, pds, ti, td0 = self._img_lists
        self._check(self.lib.dtqn_img_td_lists(self._net_ref, replay.view_ref, self._td_ref, _p(pi), _p(pd0), _p(pd1), _p(pds), _p(ti), _p(td0), s),
                    "dtqn_img_td_lists")
        self.img.prep(self.theta_pol, s)
        # the target parameters change only at a hard sync: their transposed copies are refreshed when the sync counter moved
        # (stats[10] is read lazily; refreshing every update costs 24 MB of writes, so it is simply redone each time here)
        self.img_tgt.prep(self.theta_tgt, s)
        n_pol, n_tgt = Bn * (L + 1), Bn * L
        self._img_act = self.img.act_buffer(n_pol, "train")
        self.img.encode(self.theta_pol, replay.obs, pi, n_pol, self._img_act, self.xemb, pd0, self.xemb, pd1, stream=s)
        self.img_tgt.encode(self.theta_tgt, replay.obs, ti, n_tgt, self.img_tgt.act_buffer(n_tgt, "target"), self.xemb, td0, stream=s)

    def _img_backward(self, replay: DeviceReplay, s):
        """Behind dtqn_td_backward: the encoder's data and weight gradients from dL/dx0 in the grd records, written into split 0 of
        gsplit at the parameters' offsets (dtqn_td_reduce then sums the splits like every other gradient)."""
        pi, pd0, pd1, pds, ti, td0 = self._img_lists
        self.img.backward(self.theta_pol, replay.obs, pi, self.batch * (self.net.ctx_len + 1), self._img_act, self.grd, pds, self.gsplit, stream=s)

    def forward_backward(self, replay: DeviceReplay):
        s, n, r, t = self._stream(), self._net_ref, replay.view_ref, self._td_ref
        if getattr(self, "_pipe", None) is not None and self.td.sample_in_kernel and self.img is None and self._pipe["ride"]:
            # one library call for the five launches in front of the optimizer (host time matters in the step loops)
            d, have, nxt = self._pipe_begin(replay)
            self._check(self.lib.dtqn_td_gradients_pipelined(n, r, t, nxt, 1 if have else 0, d, s), "dtqn_td_gradients_pipelined")
            return
        if self.img is not None:
            self._img_encode_windows(replay, s)
        self._forward_stage(replay, s)
        self._backward_stage(replay, s)
        if self.img is not None:
            self._img_backward(replay, s)
        self._check(self.lib.dtqn_td_wgrad(n, t, s), "dtqn_td_wgrad")
        self._check(self.lib.dtqn_td_reduce(n, t, s), "dtqn_td_reduce")

    def recompute_gradnorm(self):
        self._check(self.lib.dtqn_td_gradnorm(ctypes.byref(self.net), ctypes.byref(self.td), self._stream()), "dtqn_td_gradnorm")

    def clip_adam(self):
        self._check(self.lib.dtqn_td_clip_adam(self._net_ref, self._td_ref, self._stream()), "dtqn_td_clip_adam")
        if getattr(self, "_pipe", None) is not None:
            self._pipe["steps"] += 1

    def target_sync(self):
        if getattr(self, "_pipe", None) is not None:
            self._pipe["tgt_version"] += 1             # a target pass launched ahead read the old parameters
        self._check(self.lib.dtqn_target_sync(ctypes.byref(self.net), _p(self.theta_pol), _p(self.theta_tgt), self._stream()),
                    "dtqn_target_sync")

    def forward(self, obs: torch.Tensor, actions: Optional[torch.Tensor], target: bool = False) -> torch.Tensor:
        """DTQN.forward on [B, n, O] float32 observations (inference; no autograd graph)."""
        Bn, n = int(obs.shape[0]), int(obs.shape[1])
        q = torch.empty((Bn, n, self.actor_net.num_actions), dtype=torch.float32, device=self.device)
        theta = self.theta_tgt if target else self.theta_pol
        rc = self.lib.dtqn_forward(self._actor_net_ref, _p(theta), _p(obs), _p(actions), Bn, n, _p(q), self._stream())
        if rc == B.DEFINES["DTQN_ERR_ARG"]:
            raise AssertionError("Cannot forward, history is longer than expected.")   # dtqn.py:170-173
        self._check(rc, "dtqn_forward")
        return q

    def read_stats(self) -> dict:
        """Blocking read-back of the statistics of the last update."""
        vals = self.stats.detach().cpu().numpy()
        return dict(zip(STAT_NAMES, (float(v) for v in vals)))
