"""Agents of the DTQN hot path.  `DtqnAgent` is the reference's name (dtqn/agents/dtqn.py:15); BASELINE.json's north_star
spells it `DTQNAgent` -- both resolve to the same class.  Resolved on first access so that importing the package does not
load the gfx950 engine."""

__all__ = ["DtqnAgent", "DTQNAgent", "VectorActor"]


def __getattr__(name):
    if name in ("DtqnAgent", "DTQNAgent"):
        from .dtqn import DtqnAgent
        return DtqnAgent
    if name == "VectorActor":
        from .vector import VectorActor
        return VectorActor
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
