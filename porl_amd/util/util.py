"""Builder helpers with the reference's module structure (so state_dict keys interchange).

Mirrors the *interface* of /root/reference/util/util.py:20-56 (`Squeeze`, `mlp`,
`update_exponential_moving_average`); the modules built here only carry parameters and shapes —
device arithmetic is done by the HIP engine (porl_amd/engine.py), never by torch.nn.functional.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class Squeeze(nn.Module):
    """Marker module: drop a singleton dimension (reference util/util.py:20-26)."""

    def __init__(self, dim=None):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x.squeeze(dim=self.dim)


def mlp(dims, activation=nn.ReLU, output_activation=None, layer_norm=False, squeeze_output=False):
    """Linear [-> LayerNorm] -> act, repeated; then Linear [-> out act] [-> Squeeze(-1)].

    Same positional layout inside nn.Sequential as the reference (util/util.py:29-47), hence the same
    parameter names ('0.weight', '2.weight', ... or '0','1','3','4',... with layer_norm) and the same
    consumption of the torch RNG at construction.
    """
    if len(dims) < 2:
        raise AssertionError("MLP requires at least two dims (input and output)")
    mods = []
    for fan_in, fan_out in zip(dims[:-2], dims[1:-1]):
        mods.append(nn.Linear(fan_in, fan_out))
        if layer_norm:
            mods.append(nn.LayerNorm(fan_out))
        mods.append(activation())
    mods.append(nn.Linear(dims[-2], dims[-1]))
    if output_activation is not None:
        mods.append(output_activation())
    if squeeze_output:
        if dims[-1] != 1:
            raise AssertionError("squeeze_output needs a scalar head")
        mods.append(Squeeze(-1))
    return nn.Sequential(*mods).to(dtype=torch.float32)


def mlp_spec(seq: nn.Sequential):
    """Decode an `mlp()` Sequential into (dims, layer_norm, out_act) for the engine."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    dims = [lin[0].in_features] + [m.out_features for m in lin]
    layer_norm = any(isinstance(m, nn.LayerNorm) for m in seq)
    out_act = "tanh" if any(isinstance(m, nn.Tanh) for m in seq) else None
    return dims, layer_norm, out_act


def update_exponential_moving_average(target, source, alpha):
    """target <- (1-alpha)*target + alpha*source (reference util/util.py:54-56), stand-alone (the training step fuses
    this into the Adam sweep).  Device tensors go through the `porl_ema` kernel when they are 16-byte aligned and a
    multiple of 4 floats long (every weight matrix and hidden bias of the engines' flat groups); odd-sized leftovers
    (a scalar head bias) and CPU modules use the two torch ops of the reference."""
    from .. import engine as E
    with torch.no_grad():
        for t, s in zip(target.parameters(), source.parameters()):
            if (t.is_cuda and s.is_cuda and t.is_contiguous() and s.is_contiguous() and t.dtype == torch.float32
                    and t.numel() % 4 == 0 and t.data_ptr() % 16 == 0 and s.data_ptr() % 16 == 0):
                E.ema(t, s, alpha)
            else:
                t.mul_(1.0 - alpha).add_(s, alpha=alpha)
