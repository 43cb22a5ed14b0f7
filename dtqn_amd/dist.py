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

import os

import torch
import torch.distributed as td


def is_distributed() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def init_from_env(device_type: str = "cuda") -> tuple:
    """Initialise the default process group from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if device_type == "cuda":
            torch.cuda.set_device(local)
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
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


class DataParallel:
    """Wraps a TdEngine: update() = local gradient kernels -> all-reduce -> clip + Adam."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = td.get_world_size(group)
        engine.td.grad_scale = 1.0 / self.world          # mean over ranks, applied inside the optimizer kernel

    def broadcast_parameters(self, src: int = 0) -> None:
        e = self.engine
        for t in (e.theta_pol, e.theta_tgt, e.adam_m, e.adam_v, e.step_counter):
            td.broadcast(t, src=src, group=self.group)

    def exchange_kind(self) -> str:
        return "rccl all_reduce"

    def allreduce_gradient(self) -> None:
        td.all_reduce(self.engine.grad, op=td.ReduceOp.SUM, group=self.group)

    def update(self, replay) -> None:
        e = self.engine
        e.forward_backward(replay)          # forward x3, loss + backward, weight gradients, reduce -> e.grad (local mean)
        self.allreduce_gradient()           # the one exchange step
        e.recompute_gradnorm()              # global norm of the summed gradient (scaled by 1/world in the kernel)
        e.clip_adam()
