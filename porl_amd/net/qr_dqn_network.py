"""QR-DQN network — drop-in for /root/reference/src/porl/net/qr_dqn_network.py:9-103: `feature_extractor =
Sequential(Linear, ReLU, ...)`, `quantile_values_layer = Linear(hidden, action_size * num_quantiles)` (same state_dict
keys, same seeded initialisation); `forward(x)` -> (batch, action_size, num_quantiles), `get_mean_q_values(x)` ->
(batch, action_size).  Computes on the HIP Q-network engine of the owning trainer; there is no CPU path."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import _native as N


class QRNetwork(nn.Module):
    def __init__(self, state_size: int, action_size: int, num_quantiles: int, hidden_sizes: List[int] = [128, 128]):
        super().__init__()
        self.action_size, self.num_quantiles = action_size, num_quantiles
        layers, d = [], state_size
        for h in hidden_sizes:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.feature_extractor = nn.Sequential(*layers)
        self.quantile_values_layer = nn.Linear(d, action_size * num_quantiles)
        self._spec = (state_size, action_size * num_quantiles, list(hidden_sizes))
        self._engine, self._which = None, 0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._engine is None:
            raise N.NativeError("QRNetwork computes on the HIP engine of a QRDQNTrainer (no CPU path)")
        return self._engine.forward(x, self._which).view(-1, self.action_size, self.num_quantiles)

    def get_mean_q_values(self, x: torch.Tensor) -> torch.Tensor:
        """Mean over the quantiles: a 1/N-weighted sum, done by the fp32 GEMM entry (no torch math on the device)."""
        from .. import engine as E
        z = self.forward(x).reshape(-1, self.num_quantiles).contiguous()          # (B*A, N)
        w = torch.full((1, self.num_quantiles), 1.0 / self.num_quantiles, dtype=torch.float32, device=z.device)
        out = torch.empty(z.shape[0], 1, dtype=torch.float32, device=z.device)
        E.gemm_f32(0, z, w, z.shape[0], 1, self.num_quantiles, self.num_quantiles, self.num_quantiles, out, 1)
        return out.view(-1, self.action_size)
