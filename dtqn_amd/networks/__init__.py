"""Network factory surface of the DTQN hot path: `DTQN` (dtqn/networks/dtqn.py:12-218), resolved on first access."""

__all__ = ["DTQN"]


def __getattr__(name):
    if name == "DTQN":
        from .dtqn import DTQN
        return DTQN
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
