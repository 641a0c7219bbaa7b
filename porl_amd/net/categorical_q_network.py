"""C51 categorical Q-network — drop-in for /root/reference/src/porl/net/categorical_q_network.py:8-113: `feature =
Sequential(Linear, ReLU, ...)`, `fc = Linear(hidden, action_size * atom_size)`, `support = linspace(v_min, v_max,
atom_size)`; `forward(x)` -> log-probabilities (batch, action_size, atom_size) (log_softmax over the atoms),
`get_q_values(x)` -> expected values (batch, action_size).  Computes on the HIP engine of the owning trainer."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import _native as N


class CategoricalQNetwork(nn.Module):
    def __init__(self, state_size: int, action_size: int, atom_size: int = 51, v_min: float = -10, v_max: float = 10,
                 hidden_sizes: List[int] = [128, 128]):
        super().__init__()
        self.action_size, self.atom_size, self.v_min, self.v_max = action_size, atom_size, v_min, v_max
        self.support = torch.linspace(v_min, v_max, atom_size)
        layers, d = [], state_size
        for h in hidden_sizes:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.feature = nn.Sequential(*layers)
        self.fc = nn.Linear(d, action_size * atom_size)
        self._spec = (state_size, action_size * atom_size, list(hidden_sizes))
        self._engine, self._which = None, 0

    def _rows(self, x, mode):
        if self._engine is None:
            raise N.NativeError("CategoricalQNetwork computes on the HIP engine of a C51Trainer (no CPU path)")
        z = self._engine.forward(x, self._which).view(-1, self.atom_size)         # (B*A, atoms) logits
        out = torch.empty_like(z)
        N.check(N.lib().porl_softmax_mask(N.ptr(z), z.stride(0), z.shape[0], self.atom_size, 0.0, mode, N.ptr(out),
                                          N.current_stream_ptr(z)), "porl_softmax_mask")
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._rows(x, 2).view(-1, self.action_size, self.atom_size)

    def get_support(self, device: torch.device) -> torch.Tensor:
        return self.support.to(device)

    def get_q_values(self, x: torch.Tensor) -> torch.Tensor:
        """sum_n p_n z_n per action (categorical_q_network.py:82-113), as an fp32 GEMM against the support."""
        from .. import engine as E
        p = self._rows(x, 1)                                                       # probabilities (B*A, atoms)
        sup = self.support.to(p.device).view(1, -1).contiguous()
        out = torch.empty(p.shape[0], 1, dtype=torch.float32, device=p.device)
        E.gemm_f32(0, p, sup, p.shape[0], 1, self.atom_size, self.atom_size, self.atom_size, out, 1)
        return out.view(-1, self.action_size)
