"""Host-side training monitor with the interface of ``libdl/metrics/monitoring.py:4-66`` (``early_stopping``), so that
the experiment scripts' ``from libdl.metrics import early_stopping`` keeps working next to the GPU measures.  Pure
Python control logic: ``step(value)`` returns True when training should stop."""
import math


class early_stopping:
    """mode 'min'|'max'; an epoch counts as an improvement when it beats the best value by more than ``min_delta``
    (absolute, or percent of the best value with ``percentage=True``); stop after ``patience`` epochs without one.
    ``patience == 0`` disables stopping; a NaN value stops at once (after the first call)."""

    def __init__(self, mode="min", min_delta=0, patience=10, percentage=False):
        if mode not in ("min", "max"):
            raise ValueError("mode " + mode + " is unknown!")
        self.mode, self.min_delta, self.patience, self.percentage = mode, min_delta, patience, percentage
        self.best = None
        self.num_bad_epochs = 0

    def is_better(self, value, best):
        if self.patience == 0:
            return True
        margin = best * self.min_delta / 100 if self.percentage else self.min_delta
        return value < best - margin if self.mode == "min" else value > best + margin

    def curr_is_better(self, metrics):
        return self.is_better(metrics, self.best)

    def step(self, metrics):
        if self.patience == 0:
            return False
        if self.best is None:
            self.best = metrics
            return False
        if math.isnan(metrics):
            return True
        if self.is_better(metrics, self.best):
            self.best, self.num_bad_epochs = metrics, 0
        else:
            self.num_bad_epochs += 1
        return self.num_bad_epochs >= self.patience
