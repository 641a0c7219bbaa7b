"""Minimal stand-in for the reference's TensorBoard/CSV Logger (src/porl/utils/logger.py:10-137), which is
observability and outside the accelerated path: same method names used by the trainers, CSV only."""
from __future__ import annotations

import csv
import os


class Logger:
    def __init__(self, log_dir="logs"):
        self.log_dir = log_dir
        self._rows = []
        self.writer = None          # no tensorboard in this build

    def log_hyperparameters(self, hparams):
        self.hparams = dict(hparams)

    def log_loss(self, step, loss):
        self._rows.append((step, loss))

    def log_step(self, *a, **k):
        pass

    def log_episode(self, *a, **k):
        pass

    def close(self):
        if not self._rows:
            return
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "metrics.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["step", "loss"])
            w.writerows(self._rows)
