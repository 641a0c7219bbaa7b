"""Implicit Quantile Network — drop-in for /root/reference/src/porl/net/iqn_network.py:10-91 (the LIVE class of that file:
`IQNNetwork(state_size, action_size, embedding_dim=64, hidden_size=512)`, `forward(states, quantiles)` ->
(batch, num_quantiles, action_size), `get_quantile_embedding`), computing on one MI355X.

Every arithmetic step is a hand-written gfx950 kernel behind the C ABI: the five Linear layers run on the grouped fp32-MFMA
GEMM (`util/hip_mlp._MlpFn`: bias + ReLU in the epilogue, dgrad with the ReLU mask, wgrad as a transposed product), the
cosine features, the Hadamard product with the state features and the action gather on csrc/iqn.hpp.  autograd only
strings the pieces together (three `torch.autograd.Function`s); there is no CPU path.

`get_q_values(states, taus)` is what IQNTrainer.learn / select_action call (iqn_trainer.py:89,98,109,114); upstream's live
class does not define it (only the commented-out older class does, iqn_network.py:247-253, as a mean over K output
quantiles this class does not have).  Here it is `forward`: one value per sampled fraction and action.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _native as N
from ..util.hip_mlp import _MlpFn, _need_device


def cos_embed(quantiles: torch.Tensor, embedding_dim: int) -> torch.Tensor:
    """(B, N) fractions -> (B * N, embedding_dim) features cos(pi * i * tau), i = 1..embedding_dim (iqn_network.py:74-91)."""
    q = quantiles.contiguous().float()
    out = torch.empty(q.numel(), embedding_dim, dtype=torch.float32, device=q.device)
    N.check(N.lib().porl_iqn_cos_embed(N.ptr(q), q.numel(), embedding_dim, N.ptr(out), N.current_stream_ptr(out)),
            "porl_iqn_cos_embed")
    return out


class _Hadamard(torch.autograd.Function):
    """out[(b, n), :] = feat[b, :] * emb[(b, n), :]   (iqn_network.py:58-62: unsqueeze/expand/multiply)"""

    @staticmethod
    def forward(ctx, feat, emb, n_tau):
        feat, emb = feat.contiguous(), emb.contiguous()
        B, H = feat.shape
        out = torch.empty_like(emb)
        N.check(N.lib().porl_iqn_hadamard(N.ptr(feat), feat.stride(0), N.ptr(emb), B, n_tau, H, N.ptr(out),
                                          N.current_stream_ptr(out)), "porl_iqn_hadamard")
        ctx.n_tau = n_tau
        ctx.save_for_backward(feat, emb)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, emb = ctx.saved_tensors
        dout = dout.contiguous()
        B, H = feat.shape
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        demb = torch.empty_like(emb) if ctx.needs_input_grad[1] else None
        if dfeat is not None or demb is not None:
            N.check(N.lib().porl_iqn_hadamard_backward(N.ptr(dout), N.ptr(feat), feat.stride(0), N.ptr(emb), B, ctx.n_tau, H,
                                                       N.ptr(dfeat), N.ptr(demb), N.current_stream_ptr(dout)),
                    "porl_iqn_hadamard_backward")
        return dfeat, demb, None


class SelectAction(torch.autograd.Function):
    """z (B, N, A), actions (B,) int64 -> z[b, n, actions[b]] (B, N): the gather of iqn_trainer.py:101-103."""

    @staticmethod
    def forward(ctx, z, actions):
        z = z.contiguous()
        actions = actions.long().contiguous()
        B, n_tau, A = z.shape
        out = torch.empty(B, n_tau, dtype=torch.float32, device=z.device)
        N.check(N.lib().porl_iqn_select(N.ptr(z), N.ptr(actions), B, n_tau, A, N.ptr(out), N.current_stream_ptr(out)),
                "porl_iqn_select")
        ctx.save_for_backward(actions)
        ctx.shape = (B, n_tau, A)
        return out

    @staticmethod
    def backward(ctx, dsel):
        (actions,) = ctx.saved_tensors
        B, n_tau, A = ctx.shape
        dsel = dsel.contiguous()
        dz = torch.empty(B, n_tau, A, dtype=torch.float32, device=dsel.device)
        N.check(N.lib().porl_iqn_scatter(N.ptr(dsel), N.ptr(actions), B, n_tau, A, N.ptr(dz), N.current_stream_ptr(dz)),
                "porl_iqn_scatter")
        return dz, None


class IQNNetwork(nn.Module):
    def __init__(self, state_size, action_size, embedding_dim=64, hidden_size=512):
        super().__init__()
        self.state_size, self.action_size, self.embedding_dim = state_size, action_size, embedding_dim
        # construction order = upstream's (iqn_network.py:16-32): a seeded build draws the same initial weights
        self.feature_net = nn.Sequential(nn.Linear(state_size, hidden_size), nn.ReLU(),
                                         nn.Linear(hidden_size, hidden_size), nn.ReLU())
        self.quantile_embedding = nn.Linear(embedding_dim, hidden_size)
        self.value_net = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(), nn.Linear(hidden_size, action_size))

    def get_quantile_embedding(self, quantiles):
        b, n = quantiles.shape
        return cos_embed(quantiles, self.embedding_dim).view(b, n, self.embedding_dim)

    def forward(self, states, quantiles):
        _need_device(states)
        if states.dim() != 2 or quantiles.dim() != 2 or states.shape[0] != quantiles.shape[0]:
            raise RuntimeError(f"expected states (B, {self.state_size}) and quantiles (B, N), got {tuple(states.shape)}, "
                               f"{tuple(quantiles.shape)}")
        if states.shape[1] != self.state_size:
            raise RuntimeError(f"expected states (B, {self.state_size}), got {tuple(states.shape)}")
        B, n_tau = quantiles.shape
        f0, f1 = self.feature_net[0], self.feature_net[2]
        v0, v1 = self.value_net[0], self.value_net[2]
        qe = self.quantile_embedding
        feat = _MlpFn.apply(states.float().contiguous(), 1, f0.weight, f0.bias, f1.weight, f1.bias)    # ReLU after both
        emb = _MlpFn.apply(cos_embed(quantiles, self.embedding_dim), 0, qe.weight, qe.bias)            # no activation
        z = _MlpFn.apply(_Hadamard.apply(feat, emb, n_tau), 0, v0.weight, v0.bias, v1.weight, v1.bias)
        return z.view(B, n_tau, self.action_size)

    def get_q_values(self, states, taus):
        return self.forward(states, taus)
