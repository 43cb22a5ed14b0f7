"""Loader of the gfx950 engine (libdtqn_hip.so).  There is NO fallback: if the library is missing
or there is no ROCm device, the product path raises."""
from __future__ import annotations

import os

# torch must be imported BEFORE libdtqn_hip.so is dlopen'ed: the torch wheel bundles its own
# libamdhip64, and the engine has to bind to that same runtime instance (its streams are torch's).
# Loading the engine first would pull in /opt/rocm's copy and every launch on a torch stream fails.
import torch  # noqa: F401

from . import _binding

_LIB = None
# DTQN_HIP_LIB: an alternative BUILD of the same engine (tools/build_variant.py, A/B experiments on the GPU box)
LIB_PATH = os.environ.get("DTQN_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libdtqn_hip.so")


class EngineUnavailable(RuntimeError):
    pass


def get_lib():
    """The hipcc-built engine.  Loading it needs the HIP runtime but not a GPU; launching does."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                f"{LIB_PATH} is missing: build it with `python -m dtqn_amd.build` (or __graft_entry__.build()). "
                "dtqn_amd has no CPU / eager fallback.")
        _LIB = _binding.load_library(LIB_PATH)
    return _LIB


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise EngineUnavailable("dtqn_amd needs a ROCm device (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import ctypes
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
