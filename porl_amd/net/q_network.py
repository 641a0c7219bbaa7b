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


class DuelingQNetwork(nn.Module):
    """Dueling head (reference q_network.py:33-68): `model` = the hidden Linear/ReLU chain, `value` = Linear(64, 1),
    `advantage` = Linear(64, A) on its 64 features, q = v + (a - mean_a a).  Same attribute names, registration order
    (value, advantage, model: the state_dict order) and RNG consumption (hidden layers first) as the reference.

    q is LINEAR in the two heads' parameters: q_j = (w_v + W_a[j] - mean_k W_a[k]) . f + (b_v + b_a[j] - mean_k b_a[k]), so
    the engine runs a plain network whose output layer is that composed layer (porl_amd/train/cql_trainer.py:
    _DuelingHeads keeps the true head parameters, composes after every change and maps the output layer's gradient back;
    both maps are products with a constant (A, A+1) matrix on the fp32-MFMA GEMM)."""

    def __init__(self, state_size, action_size, hidden_sizes=[64, 128, 64]):
        super().__init__()
        layers, cur = [], state_size
        for h in hidden_sizes:
            layers += [nn.Linear(cur, h), nn.ReLU()]
            cur = h
        if cur != 64:
            raise ValueError("the reference's dueling heads are Linear(64, .): the last hidden size must be 64")
        self.value = nn.Sequential(nn.Linear(64, 1))
        self.advantage = nn.Sequential(nn.Linear(64, action_size))
        self.model = nn.Sequential(*layers)
        self._spec = (state_size, action_size, list(hidden_sizes))
        self._engine, self._which = None, 0

    def forward(self, x):
        """x: (b, s) -> q values (b, a)."""
        if self._engine is None:
            from .._native import NativeError
            raise NativeError("DuelingQNetwork computes on the HIP engine of a trainer (no CPU path)")
        return self._engine.forward(x, self._which)
