"""Q-network with the reference's module structure (/root/reference/src/porl/net/q_network.py:8-30):
`model = Sequential(Linear, ReLU, ..., Linear)`, hidden sizes [64, 128, 64] by default, so state_dict keys
(`model.0.weight`, ...) and seeded initialisation match.  `forward` runs on the HIP engine the module is
attached to (porl_amd/train/cql_trainer.py)."""
from __future__ import annotations

import torch
import torch.nn as nn


class QNetwork(nn.Module):
    def __init__(self, state_size, action_size, hidden_sizes=[64, 128, 64]):
        super().__init__()
        layers, cur = [], state_size
        for h in hidden_sizes:
            layers += [nn.Linear(cur, h), nn.ReLU()]
            cur = h
        layers.append(nn.Linear(cur, action_size))
        self.model = nn.Sequential(*layers)
        self._spec = (state_size, action_size, list(hidden_sizes))
        self._engine, self._which = None, 0

    def forward(self, x):
        """x: (b, s) -> q values (b, a)."""
        if self._engine is None:
            from .._native import NativeError
            raise NativeError("QNetwork computes on the HIP engine of a trainer (no CPU path); "
                              "build it through porl_amd.train.cql_trainer.CQLTrainer")
        return self._engine.forward(x, self._which)
