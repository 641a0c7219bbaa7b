"""Behaviour-policy model pi_b(a|s) of discrete BCQ — drop-in for /root/reference/src/porl/net/behavior_policy.py:9-55:
`network = Sequential(Linear, ReLU, Linear, ReLU, Linear)` with hidden sizes [64, 128] (same state_dict keys and seeded
initialisation), `forward(state)` -> action probabilities, `sample(state, threshold)` -> 0/1 mask of the actions whose
probability exceeds the threshold.  The logits come from the HIP Q-network engine the owning trainer attaches
(`porl_qnet_forward`), softmax / thresholding from `porl_softmax_mask`; there is no CPU path."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import _native as N


class BehaviorPolicy(nn.Module):
    def __init__(self, state_size: int, action_size: int, hidden_sizes: List[int] = [64, 128]):
        super().__init__()
        self.action_size = action_size
        self.network = nn.Sequential(
            nn.Linear(state_size, hidden_sizes[0]), nn.ReLU(),
            nn.Linear(hidden_sizes[0], hidden_sizes[1]), nn.ReLU(),
            nn.Linear(hidden_sizes[1], action_size),
        )
        self._spec = (state_size, action_size, list(hidden_sizes))
        self._engine = None

    def logits(self, state):
        if self._engine is None:
            raise N.NativeError("BehaviorPolicy computes on the HIP engine of a BCQTrainer (no CPU path)")
        return self._engine.forward(state, 0)

    def _softmax(self, state, threshold, write_probs):
        z = self.logits(state)
        out = torch.empty_like(z)
        N.check(N.lib().porl_softmax_mask(N.ptr(z), z.stride(0), z.shape[0], self.action_size, float(threshold),
                                          int(write_probs), N.ptr(out), N.current_stream_ptr(z)), "porl_softmax_mask")
        return out

    def forward(self, state: torch.Tensor) -> torch.Tensor:
        """(b, s) -> (b, a) normalised probabilities over the actions."""
        return self._softmax(state, 0.0, True)

    def sample(self, state: torch.Tensor, threshold: float = 0.1) -> torch.Tensor:
        """(b, s) -> (b, a) fp32 mask: 1 where pi_b(a|s) > threshold."""
        return self._softmax(state, threshold, False)
