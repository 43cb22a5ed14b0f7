"""Running statistics and the CSV logger (reference: utils/logging_utils.py).  The CSV files keep
the reference's headers and column order so existing plotting scripts keep working."""
import csv
import os
from collections import deque
from datetime import datetime
from typing import Callable, Dict, Optional


class RunningAverage:
    """Mean of the last `size` values (utils/logging_utils.py:10-24)."""

    def __init__(self, size: int):
        self.size = size
        self.q = deque()
        self.sum = 0

    def add(self, val) -> None:
        self.q.append(val)
        self.sum += val
        if len(self.q) > self.size:
            self.sum -= self.q.popleft()

    def mean(self):
        return self.sum / max(len(self.q), 1)


class DeferredRunningAverage(RunningAverage):
    """RunningAverage whose producer is asynchronous: the learner's statistics come back from the
    GPU through a pinned ring, and are only forced to completion when somebody asks for the mean
    (run.py reads it every eval_frequency steps) instead of blocking every update on `.item()`."""

    def __init__(self, size: int, drain: Optional[Callable[[], None]] = None):
        super().__init__(size)
        self._drain = drain

    def mean(self):
        if self._drain is not None:
            self._drain()
        return super().mean()


def timestamp() -> str:
    return datetime.now().strftime("%B %d, %H:%M:%S")


LOSS_COLUMNS = [("TD Error", "losses/TD_Error"), ("Grad Norm", "losses/Grad_Norm"),
                ("Max Q Value", "losses/Max_Q_Value"), ("Mean Q Value", "losses/Mean_Q_Value"),
                ("Min Q Value", "losses/Min_Q_Value"), ("Max Target Value", "losses/Max_Target_Value"),
                ("Mean Target Value", "losses/Mean_Target_Value"), ("Min Target Value", "losses/Min_Target_Value")]


class CSVLogger:
    """`<path>_results.csv` and `<path>_losses.csv`, appended to if they exist (resume)."""

    def __init__(self, path: str, args):
        self.results_path, self.losses_path = path + "_results.csv", path + "_losses.csv"
        self.envs = list(args.envs)
        if not os.path.exists(self.results_path):
            header = ["Hours", "Step"]
            for env in self.envs:
                header += [f"{env}/SuccessRate", f"{env}/EpisodeLength", f"{env}/Return"]
            self._append(self.results_path, header, mode="w")
        if not os.path.exists(self.losses_path):
            self._append(self.losses_path, ["Hours", "Step"] + [c for c, _ in LOSS_COLUMNS], mode="w")

    @staticmethod
    def _append(path, row, mode="a"):
        with open(path, mode) as f:
            csv.writer(f).writerow(row)

    def log(self, results: Dict[str, float], step: int) -> None:
        row = [results["losses/hours"], step]
        for env in self.envs:
            row += [results[f"{env}/SuccessRate"], results[f"{env}/EpisodeLength"], results[f"{env}/Return"]]
        self._append(self.results_path, row)
        self._append(self.losses_path, [results["losses/hours"], step] + [results[k] for _, k in LOSS_COLUMNS])


def get_logger(policy_path: str, args, wandb_kwargs: Dict[str, str]):
    """CSV logger, or wandb when it is installed and not disabled (it is absent from this image)."""
    if not args.disable_wandb:
        try:
            import wandb
        except ImportError:
            print("wandb is not installed: logging to CSV instead (pass --disable-wandb to silence this).")
        else:
            keys = ["model", "obs_embed", "a_embed", "in_embed", "context", "layers", "bag_size", "gate", "identity",
                    "history", "pos"]
            cfg = vars(args)
            wandb.init(project=cfg["project_name"], group="_".join(f"{k}={cfg[k]}" for k in cfg if k in keys),
                       config=cfg, **wandb_kwargs)
            return wandb
    return CSVLogger(policy_path, args)
