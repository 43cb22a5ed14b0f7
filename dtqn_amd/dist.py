"""Data-parallel replication of the learner: one process per GPU, torch.distributed backend
"nccl" (= RCCL over xGMI on MI355X), launched with torch.distributed.run.

The reference is single-process; this is new.  Every rank owns its env(s),
its device replay and its RNG streams (seed + rank), samples a local batch, and holds identical
parameters and Adam state.  Per update there is exactly ONE collective: an all-reduce (sum) of the
flat gradient (0.43 MB at cfg 1) between the gradient kernels and the clip+Adam kernel; the mean
over ranks, the global-norm clip and Adam then run identically everywhere, which equals a single
learner on the union batch because the loss is a mean over equally sized shards.
"""
from __future__ import annotations

import logging
import os

import torch
import torch.distributed as td

log = logging.getLogger("dtqn_amd.dist")


def _inject(what: str, rank: int) -> bool:
    """Fault injection for the start-up checks of the gradient exchange (tests only): DTQN_DP_INJECT=<what>:<rank>[,<what>:<rank>...],
    what = mapping | sum | local -- `mapping`: this rank fails to map its peers' buffers; `sum`: its device-side exchange returns one wrong
    element; `local`: its local check work raises.  The point of all three: EVERY rank must then land on the collective together."""
    spec = os.environ.get("DTQN_DP_INJECT", "")
    return any(tok.strip() == f"{what}:{rank}" for tok in spec.split(",") if tok)


def peer_access_row(device) -> list:
    """can_device_access_peer(this device, d) for every visible device d (1 on the diagonal): one row of the matrix bench.py prints."""
    if torch.device(device).type != "cuda":
        return []
    dev = torch.device(device)
    mine = dev.index if dev.index is not None else torch.cuda.current_device()
    return [1 if d == mine else int(torch.cuda.can_device_access_peer(mine, d)) for d in range(torch.cuda.device_count())]


def is_distributed() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def same_device() -> bool:
    """DTQN_DIST_SAME_DEVICE=1: every rank uses cuda:0 and gloo carries the collectives -- what a ONE-GPU box can run of the
    multi-rank path (RCCL refuses two ranks on one GPU).  For smoke tests of the launch / timing / reporting code, not for numbers."""
    return os.environ.get("DTQN_DIST_SAME_DEVICE", "0") == "1"


def init_from_env(device_type: str = "cuda") -> tuple:
    """Initialise the default process group from torchrun's environment.  Returns (rank, world, local device index)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if same_device() else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if device_type == "cuda" and not same_device():
            torch.cuda.set_device(local)
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            if device_type == "cuda":
                torch.cuda.set_device(local)
            td.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def rank() -> int:
    return td.get_rank() if is_distributed() else 0


def agree_all(flag: bool, device) -> bool:
    """True iff `flag` is true on EVERY rank (one tiny all-reduce + a host read: use off the per-update path)."""
    if not is_distributed():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    td.all_reduce(t, op=td.ReduceOp.MIN)
    return bool(t.item())


def agree_any(flag: bool, device) -> bool:
    """True iff `flag` is true on ANY rank."""
    if not is_distributed():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return bool(t.item())


class P2PExchange:
    """Device-side gradient exchange (include/dtqn_hip.h, dtqn_td_xreduce): every rank exports one allocation
    [gx generation 0 | gx generation 1 | flag word] to its peers and maps theirs; an update then needs no library call --
    ONE launch that raises the rank's flag and reads all peers directly (7 xGMI links per GPU on an MI355X node: a
    rank reads its 7 peers at once).  The handles travel through torch's own IPC reductions (hipIpcGetMemHandle over dmabuf:
    HSA_ENABLE_IPC_MODE_LEGACY=0) and `all_gather_object`; on a CPU test build the buffer is POSIX shared memory."""

    def __init__(self, engine, group=None):
        """Collective: every rank of `group` must construct one.  The ONLY collective inside is one all_gather_object that every rank
        reaches whatever happens locally; a rank whose export or mapping fails records it in `self.error` (the caller votes on it,
        DataParallel.__init__) instead of leaving the others alone in a later collective."""
        from torch.multiprocessing import reductions  # noqa: F401  (registers the reducers with ForkingPickler)
        self.engine, self.group = engine, group
        self.world, self.rank = td.get_world_size(group), td.get_rank(group)
        n = self.n = engine.net.n_trainable
        dev = engine.device
        self.error = None
        self.buf = torch.zeros(2 * n + 16, dtype=torch.float32, device=dev)
        # ForkingPickler carries torch's IPC reducers (a storage travels as its shared-memory / IPC handle, not as a copy of its bytes)
        import pickle
        from multiprocessing.reduction import ForkingPickler
        payload = None
        prev_strategy = None
        try:
            if dev.type != "cuda":
                # handles that survive pickling through a collective; the process-wide setting is put back below
                prev_strategy = torch.multiprocessing.get_sharing_strategy()
                torch.multiprocessing.set_sharing_strategy("file_system")
                self.buf.share_memory_()
            payload = bytes(ForkingPickler.dumps(self.buf))
        except Exception as exc:
            self.error = f"export: {type(exc).__name__}: {exc}"[:200]
        finally:
            if prev_strategy is not None:
                torch.multiprocessing.set_sharing_strategy(prev_strategy)
        handles = [None] * self.world
        td.all_gather_object(handles, payload, group=group)
        self.peers = [self.buf] * self.world
        try:
            if self.error is None and any(h is None for h in handles):
                raise RuntimeError("a peer could not export its exchange buffer")
            if self.error is None and _inject("mapping", self.rank):
                raise RuntimeError("injected mapping failure (DTQN_DP_INJECT)")
            if self.error is None:
                self.peers = [self.buf if r == self.rank else pickle.loads(h) for r, h in enumerate(handles)]
                if dev.type == "cuda":
                    for r, p in enumerate(self.peers):          # a first copy makes torch enable peer access to that device
                        if r != self.rank and p.device != dev:
                            # a kernel that reads a buffer its device cannot reach faults the whole process: ask first, and let
                            # the vote send every rank to the collective instead
                            mine = dev.index if dev.index is not None else torch.cuda.current_device()
                            if not torch.cuda.can_device_access_peer(mine, p.device.index):
                                raise RuntimeError(f"device {mine} has no peer access to device {p.device.index}")
                            # torch enables peer access lazily, per direction, inside its device-to-device copies -- for the SOURCE
                            # device of the copy towards the destination's memory (the copy kernel runs on the source).  The reduce
                            # kernel runs HERE and reads THERE: the copy that enables that direction goes from this device to the peer's
                            # (the other one is made too, so that either runtime convention is covered)
                            self.buf[2 * n + 8:2 * n + 9].to(p.device)
                            p[2 * n + 8:2 * n + 9].to(dev)
        except Exception as exc:
            self.error = f"mapping: {type(exc).__name__}: {exc}"[:200]
            self.peers = [self.buf] * self.world
        esz = 4
        ptrs = lambda off: torch.tensor([p.data_ptr() + off * esz for p in self.peers], dtype=torch.int64, device=dev)
        self.grad_ptrs = [ptrs(0), ptrs(n)]
        self.flag_ptrs = ptrs(2 * n)
        self.own = [self.buf[:n], self.buf[n:2 * n]]
        self.own_flag = self.buf[2 * n:2 * n + 1]
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        # the optimizer kernel reads the status word: an exchange that gave up skips the update (and every later one) instead of
        # applying a stale sum, and the host raises when it drains that update's statistics (DtqnAgent._drain_stats)
        engine.td.xstatus = self.status.data_ptr()
        self.k = 0
        # (no barrier here: the caller's vote on `error` is the point every rank has mapped every buffer before anyone publishes)

    def begin(self) -> None:
        """Point the gradient kernels of the next update at this generation's exchange buffer."""
        self.k += 1
        self.engine.td.grad = self.own[self.k & 1].data_ptr()

    def reduce(self) -> None:
        import ctypes
        e = self.engine
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        s = e._stream()
        # one launch: it raises this rank's flag (the kernel boundary in front of it has made the gradient visible), then waits and sums
        e._check(e.lib.dtqn_td_xreduce(e._net_ref, e._td_ref, vp(self.grad_ptrs[self.k & 1]), vp(self.flag_ptrs), self.world, self.k,
                                       vp(e.grad), vp(self.status), vp(self.own_flag), s), "dtqn_td_xreduce")
        e.td.grad = e.grad.data_ptr()

    def check(self) -> None:
        """Raise if a wait inside dtqn_td_xreduce ran out (a peer died or skipped an update).  Synchronises."""
        if int(self.status.item()) != 0:
            raise RuntimeError("device-side gradient exchange: a peer's gradient never arrived (bounded wait expired)")


class DataParallel:
    """Wraps a TdEngine: update() = local gradient kernels -> exchange (sum over ranks) -> clip + Adam.

    Two exchanges exist: `rccl` = torch.distributed.all_reduce + a norm-recompute launch, and `p2p` = the device-side exchange
    (P2PExchange: no library call, one reduce launch reading every peer directly).  DTQN_DP_EXCHANGE / `exchange` = rccl | p2p |
    auto; the default is AUTO: map the peers, run the device-side exchange on known vectors for both buffer generations,
    compare with the sums an all_reduce gives, and keep it only if EVERY rank got every element right -- otherwise (mapping
    failed, a wait timed out, a sum is off) all ranks fall back to the collective together.  `selection` records what happened
    (bench.py prints it): {"kind": ..., "validated": ..., "reason": ...}."""

    def __init__(self, engine, group=None, exchange=None):
        self.engine = engine
        self.group = group
        self.world = td.get_world_size(group)
        engine.td.grad_scale = 1.0 / self.world          # mean over ranks, applied inside the optimizer kernel
        kind = (exchange or os.environ.get("DTQN_DP_EXCHANGE", "auto")).lower()
        if kind not in ("rccl", "p2p", "auto"):
            raise ValueError("DTQN_DP_EXCHANGE must be 'rccl', 'p2p' or 'auto'")
        self.p2p = None
        self.selection = {"kind": "rccl", "validated": False, "reason": "requested"}
        if kind == "rccl":
            return
        self.p2p = P2PExchange(engine, group)             # one collective inside, reached by every rank; failures land in .error
        err = self.p2p.error
        mapped = agree_all(err is None, self._vote_device())
        if not mapped:
            self._drop_p2p()             # (also in front of the raise: engine.td.xstatus must not keep pointing at a discarded buffer)
            if kind == "p2p":
                raise RuntimeError(f"DTQN_DP_EXCHANGE=p2p but the peers' exchange buffers could not be mapped on every rank ({err})")
            self.selection = {"kind": "rccl", "validated": False, "reason": f"peer mapping failed on some rank ({err or 'another rank'})"}
            self._log_selection()
            return
        ok, why = self._validate_p2p()
        if ok:
            self.selection = {"kind": "p2p", "validated": True, "reason": why}
        else:
            self._drop_p2p()
            if kind == "p2p":
                raise RuntimeError(f"DTQN_DP_EXCHANGE=p2p failed its start-up check: {why}")
            self.selection = {"kind": "rccl", "validated": False, "reason": f"device-side exchange failed its start-up check: {why}"}
        self._log_selection()

    def _log_selection(self) -> None:
        """Once, on rank 0: which exchange the job runs and why (a fall-back to the collective is a warning: it is slower, and the only
        other place that says so is bench.py's line)."""
        if td.get_rank(self.group) != 0:
            return
        sel = self.selection
        if sel["kind"] == "rccl" and sel["reason"] != "requested":
            log.warning("gradient exchange: falling back to the all_reduce collective -- %s", sel["reason"])
        else:
            log.info("gradient exchange: %s (%s)", sel["kind"], sel["reason"])

    def _vote_device(self):
        # same-device smoke mode runs the collectives over gloo (host tensors)
        return "cpu" if (same_device() or self.engine.device.type != "cuda") else self.engine.device

    def _drop_p2p(self) -> None:
        self.p2p = None
        self.engine.td.xstatus = None
        self.engine.td.grad = self.engine.grad.data_ptr()

    def _validate_p2p(self):
        """Two exchanges (one per buffer generation) of vectors whose sums are exact in any order: rank r contributes
        (r + 1) * (i mod 7 + g) -- small integers --, so the rank-ordered device-side sum must EQUAL the collective's.
        Every rank issues the SAME collectives whatever happens locally: exceptions are caught around the local work only (copy,
        reduce launch, compare), the all_reduce of each generation and the closing vote are unconditional."""
        e, p = self.engine, self.p2p
        n = p.n
        # a start-up check must not sit out the 5 s of a training run: 2 s through the DtqnTd field (per engine; the process
        # environment is left alone), unless the user set a bound of their own
        user_bound = "DTQN_XCH_TIMEOUT_MS" in os.environ
        if not user_bound:
            e.td.xch_timeout_ms = 2000
        ok, why = True, "two generations equal to all_reduce on every rank"
        host_side = same_device() or e.device.type != "cuda"
        base = torch.arange(n, dtype=torch.float32, device=e.device) % 7
        for g in (1, 2):
            vec = (base + g) * float(p.rank + 1)
            got = None
            try:
                if _inject("local", p.rank):
                    raise RuntimeError("injected local failure (DTQN_DP_INJECT)")
                p.k += 1
                p.own[p.k & 1].copy_(vec)
                p.reduce()
                if _inject("sum", p.rank):
                    e.grad[5] += 1.0
                got = e.grad.cpu() if host_side else e.grad.clone()
            except Exception as exc:            # local: this rank still walks through the collectives below
                if ok:
                    ok, why = False, f"generation {g}: {type(exc).__name__}: {exc}"[:200]
            want = vec.cpu() if host_side else vec.clone()
            td.all_reduce(want, op=td.ReduceOp.SUM, group=self.group)         # unconditional, every generation, every rank
            if not ok or got is None:
                continue
            try:
                if int(p.status.item()) != 0:
                    ok, why = False, f"generation {g}: a peer's flag never arrived (bounded wait expired)"
                elif not torch.equal(got, want):
                    bad = int((got != want).sum().item())
                    ok, why = False, f"generation {g}: {bad} of {n} sums differ from all_reduce"
                else:
                    norm_got = float(e.norm_partial.sum().item())
                    norm_want = float((want.double() ** 2).sum().item())
                    if abs(norm_got - norm_want) > 1e-5 * norm_want:
                        ok, why = False, f"generation {g}: sum of squares {norm_got} != {norm_want}"
            except Exception as exc:
                ok, why = False, f"generation {g}: {type(exc).__name__}: {exc}"[:200]
        if not user_bound:
            e.td.xch_timeout_ms = 0
        all_ok = agree_all(ok, self._vote_device())
        if not all_ok and ok:
            why = "another rank's check failed"
        # the gradient kernels never write the padding between tensors and rely on it being zero: wipe the test vectors (a failed
        # wait leaves the status word set: those buffers are dropped together with the exchange)
        e.grad.zero_()
        e.norm_partial.zero_()
        if all_ok:
            p.own[0].zero_()
            p.own[1].zero_()
            if e.device.type == "cuda":
                torch.cuda.synchronize(e.device)
        td.barrier(group=self.group)               # nobody publishes a real gradient before every buffer is clean
        return all_ok, why

    def broadcast_parameters(self, src: int = 0) -> None:
        e = self.engine
        for t in (e.theta_pol, e.theta_tgt, e.adam_m, e.adam_v, e.step_counter):
            td.broadcast(t, src=src, group=self.group)

    def exchange_kind(self) -> str:
        return "device-side p2p reduce (dtqn_td_xreduce)" if self.p2p is not None else "rccl all_reduce + dtqn_td_gradnorm"

    def time_exchange(self, which: str, iters: int = 50, warm: int = 10):
        """HIP-event timing of ONE exchange step by itself (every rank must call it): `which` = "selected" | "rccl".
        Returns per-iteration microseconds (list) or None when that exchange is not available."""
        e = self.engine
        if e.device.type != "cuda":
            return None
        stream = e._bound_torch_stream or torch.cuda.current_stream(e.device)
        ts = []
        with torch.cuda.stream(stream):
            for i in range(warm + iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                if which == "rccl":
                    if same_device():
                        return None               # gloo moves host tensors: not an exchange worth timing
                    td.all_reduce(e.grad, op=td.ReduceOp.SUM, group=self.group)
                    e.recompute_gradnorm()
                else:
                    self.allreduce_gradient()
                e1.record(stream)
                e1.synchronize()
                if i >= warm:
                    ts.append(e0.elapsed_time(e1) * 1e3)
        return ts

    def allreduce_gradient(self) -> None:
        """The exchange step by itself (bench.py times it): e.grad <- sum over ranks, norm partials of the sum."""
        if self.p2p is not None:
            self.p2p.k += 1                 # a generation of its own: what it exchanges is whatever the buffer holds
            self.p2p.reduce()
            return
        td.all_reduce(self.engine.grad, op=td.ReduceOp.SUM, group=self.group)
        self.engine.recompute_gradnorm()

    def forward_backward(self, replay) -> None:
        """forward x3, loss + backward, weight gradients, reduce -> the local mean gradient, where the exchange reads it."""
        if self.p2p is not None:
            self.p2p.begin()
        self.engine.forward_backward(replay)

    def reduce(self) -> None:
        """The one exchange step: after it e.grad holds the sum over ranks and norm_partial its sums of squares."""
        if self.p2p is not None:
            self.p2p.reduce()
        else:
            td.all_reduce(self.engine.grad, op=td.ReduceOp.SUM, group=self.group)
            self.engine.recompute_gradnorm()              # global norm of the summed gradient (scaled by 1/world in the kernel)

    def update(self, replay) -> None:
        self.forward_backward(replay)
        self.reduce()
        self.engine.clip_adam()
