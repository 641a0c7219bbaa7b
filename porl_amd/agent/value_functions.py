"""Value heads with the reference's interface (/root/reference/agent/value_functions.py:6-42).

The modules only hold parameters (same names/shapes/initialisation as the reference); `both()` /
`forward()` run on the HIP engine the module is attached to.  A module that is not part of an agent
attaches itself to a private engine on first use.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..util.util import mlp


class _EngineBacked(nn.Module):
    """Mixin: lazily create a private IqlEngine for stand-alone use."""

    _engine = None          # set by the owning agent (shared) or created lazily (private)
    _engine_role = None     # 'vf' | 'target' | 'policy'
    _private = False        # True when the engine was created by this module for stand-alone use

    def _private_engine(self, obs_dim, pol_out_dim, hidden_dim, n_hidden, layer_norm, pol_tanh, batch):
        from ..engine import IqlEngine
        dev = next(self.parameters()).device
        eng = IqlEngine(obs_dim, pol_out_dim, hidden_dim, n_hidden, layer_norm, pol_tanh, 0, max(batch, 256), dev)
        return eng


class TwinV(_EngineBacked):
    def __init__(self, state_dim, layer_norm=False, hidden_dim=256, n_hidden=2):
        super().__init__()
        dims = [state_dim, *([hidden_dim] * n_hidden), 1]
        self.v1 = mlp(dims, layer_norm=layer_norm, squeeze_output=True)
        self.v2 = mlp(dims, layer_norm=layer_norm, squeeze_output=True)
        self._spec = (state_dim, hidden_dim, n_hidden, bool(layer_norm))

    def _attach_private(self, batch):
        from ..engine import IqlEngine
        S, H, L, ln = self._spec
        eng = self._private_engine(S, 1, H, L, ln, False, batch)
        views = IqlEngine.views(eng.params_vf, eng.tensor_table(IqlEngine.GROUP_VF))
        with torch.no_grad():
            for p, v in zip(self.parameters(), views):
                v.copy_(p)
                p.data = v
        self._engine, self._engine_role, self._private = eng, "vf", True

    def both(self, state):
        if self._engine is None or (self._private and self._engine.cfg.max_batch < state.shape[0]):
            self._attach_private(state.shape[0])
        return self._engine.forward_value(state, target=self._engine_role == "target")

    def forward(self, state):
        return torch.min(*self.both(state))


class TwinQ(nn.Module):
    """Twin Q(s, a) heads (reference value_functions.py:6-18; unused by POR / SORL).  `both` / `forward` run the two
    Linear/ReLU chains on the device through the fp32-MFMA GEMM (util/hip_mlp.py), differentiable like any module."""

    def __init__(self, state_dim, action_dim, hidden_dim=256, n_hidden=2):
        super().__init__()
        dims = [state_dim + action_dim, *([hidden_dim] * n_hidden), 1]
        self.q1 = mlp(dims, squeeze_output=True)
        self.q2 = mlp(dims, squeeze_output=True)

    def both(self, state, action):
        from ..util.hip_mlp import mlp_forward
        sa = torch.cat([state, action], 1)
        return mlp_forward(self.q1, sa), mlp_forward(self.q2, sa)

    def forward(self, state, action):
        return torch.min(*self.both(state, action))


class ValueFunction(nn.Module):
    """Single V(s) head (reference value_functions.py:21-28; unused by POR / SORL); forward on the device through
    util/hip_mlp.py."""

    def __init__(self, state_dim, hidden_dim=256, n_hidden=2):
        super().__init__()
        self.v = mlp([state_dim, *([hidden_dim] * n_hidden), 1], squeeze_output=True)

    def forward(self, state):
        from ..util.hip_mlp import mlp_forward
        return mlp_forward(self.v, state)
