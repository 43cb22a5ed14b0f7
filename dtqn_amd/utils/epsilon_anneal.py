"""Exploration schedules (reference: utils/epsilon_anneal.py)."""


class Constant:
    def __init__(self, start: float):
        self.val = start

    def anneal(self) -> None:
        return None


class LinearAnneal:
    """Despite the name the decay is geometric towards `end`: each call removes 1/duration of the
    remaining distance (utils/epsilon_anneal.py:33-34).  Reproduced as is -- it defines the
    exploration schedule the published results were obtained with."""

    def __init__(self, start: float, end: float, duration: int):
        self.val, self.min, self.duration = start, end, duration

    def anneal(self) -> None:
        self.val = max(self.min, self.val - (self.val - self.min) / self.duration)
