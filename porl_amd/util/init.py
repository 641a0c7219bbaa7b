"""Seeded parameter construction in the reference's creation order.

POR builds goal_policy before vf (/root/reference/agent/por.py:36-45); SORL builds v_net before
policy (/root/reference/agent/sorl.py:37-45).  torch's default nn.Linear init draws from the global
torch RNG, so the same seed + the same order gives the reference's initial weights.  Pure host code.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .util import mlp


def _mlp_sd(prefix, seq):
    return OrderedDict((f"{prefix}.{k}", v.detach().numpy().copy()) for k, v in seq.state_dict().items())


def build_por_state_dict(S, H, L, layer_norm=False, seed=0, policy_out=None):
    """Returns an OrderedDict in POR.state_dict() key order (31 tensors for L=2, no LayerNorm)."""
    torch.manual_seed(seed)
    out_dim = S if policy_out is None else policy_out
    pol = mlp([S, *([H] * L), out_dim])
    v1 = mlp([S, *([H] * L), 1], layer_norm=layer_norm, squeeze_output=True)
    v2 = mlp([S, *([H] * L), 1], layer_norm=layer_norm, squeeze_output=True)
    sd = OrderedDict()
    sd["goal_policy.log_std"] = np.zeros(out_dim, dtype=np.float32)
    sd.update(_mlp_sd("goal_policy.net", pol))
    sd.update(_mlp_sd("vf.v1", v1))
    sd.update(_mlp_sd("vf.v2", v2))
    for k in [k for k in sd if k.startswith("vf.")]:
        sd["v_target." + k[3:]] = sd[k].copy()
    return sd


def build_sorl_state_dict(S, A, H, L, layer_norm=False, seed=0):
    """SORL.state_dict() key order: v_net.*, policy.log_std, policy.net.*, v_tgt.*."""
    torch.manual_seed(seed)
    v1 = mlp([S, *([H] * L), 1], layer_norm=layer_norm, squeeze_output=True)
    v2 = mlp([S, *([H] * L), 1], layer_norm=layer_norm, squeeze_output=True)
    pol = mlp([S, *([H] * L), A], output_activation=torch.nn.Tanh)
    sd = OrderedDict()
    sd.update(_mlp_sd("v_net.v1", v1))
    sd.update(_mlp_sd("v_net.v2", v2))
    sd["policy.log_std"] = np.zeros(A, dtype=np.float32)
    sd.update(_mlp_sd("policy.net", pol))
    for k in [k for k in sd if k.startswith("v_net.")]:
        sd["v_tgt." + k[6:]] = sd[k].copy()
    return sd
