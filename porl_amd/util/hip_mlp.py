"""A Linear/ReLU(/Tanh) chain on the device through the fp32-MFMA GEMM of the C ABI (`porl_gemm_f32`), WITH autograd.

The agents' hot path never comes through here: POR / SORL / CQL run their fused update engines.  This is the forward
(and backward) of the stand-alone surface classes the reference defines but no training script calls —
`TwinQ`, `ValueFunction`, `DeterministicPolicy` (agent/value_functions.py:6-28, agent/policy.py:62-73) — and of
`GaussianPolicy.act(..., enable_grad=True)` (agent/policy.py:30-33).  Every product is a hand-written gfx950 kernel
launch (forward: bias + ReLU/tanh in the epilogue; backward: dgrad with the ReLU mask in the epilogue, wgrad as a
transposed product); the only torch tensor ops are the tanh derivative on a (B, out) tensor and the bias gradients'
column sums.  There is no CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _native as N
from .. import engine as E
from .util import Squeeze


def _need_device(x):
    if x.device.type != "cuda":
        raise N.NativeError("porl_amd computes on a HIP device only (tensor on %s); there is no CPU path" % x.device)


def _linear(x, w, b, act):
    """act(x w^T + b) with act 0 none / 1 relu / 2 tanh; x (B, K) row-major with any row stride, w (N, K) likewise."""
    B, K = x.shape
    n = w.shape[0]
    y = torch.empty(B, n, dtype=torch.float32, device=x.device)
    E.gemm_f32(0, x, w, B, n, K, x.stride(0), w.stride(0), y, n, bias=b, act=act)
    return y


def _rows(t):
    """Unit column stride and a row stride torch reports faithfully (a (B, 1) view may report anything)."""
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.shape[0] > 1 and t.shape[1] > 1) else t.contiguous()


class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_act, *params):
        n_lin = len(params) // 2
        acts = [x]
        h = x
        for l in range(n_lin):
            w, b = params[2 * l], params[2 * l + 1]
            h = _linear(h, _rows(w), b.contiguous(), 1 if l < n_lin - 1 else out_act)
            acts.append(h)
        ctx.out_act, ctx.n_lin = out_act, n_lin
        ctx.save_for_backward(*acts, *params)
        return h

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        n_lin = ctx.n_lin
        acts, params = saved[:n_lin + 1], saved[n_lin + 1:]
        dz = dy.contiguous()
        if ctx.out_act == 2:
            dz = dz * (1.0 - acts[-1] * acts[-1])                    # tanh'(z) from the stored output
        elif ctx.out_act == 1:
            dz = dz * (acts[-1] > 0).to(dz.dtype)                    # a chain that ends in ReLU (IQN's feature_net)
        grads = [None] * (2 * n_lin)
        dx = None
        for l in range(n_lin - 1, -1, -1):
            w = _rows(params[2 * l])
            a_in = acts[l]
            B, K = a_in.shape
            n = w.shape[0]
            if ctx.needs_input_grad[2 + 2 * l]:
                dw = torch.empty(n, K, dtype=torch.float32, device=dz.device)
                # wgrad dW = dZ^T A: "TN" mode, the batch is the contraction
                E.gemm_f32(2, dz, a_in, n, K, B, dz.stride(0), a_in.stride(0), dw, K)
                grads[2 * l] = dw.view_as(params[2 * l])
            if ctx.needs_input_grad[3 + 2 * l]:
                grads[2 * l + 1] = dz.sum(0)
            if l > 0 or ctx.needs_input_grad[0]:
                dprev = torch.empty(B, K, dtype=torch.float32, device=dz.device)
                # dgrad dA = dZ W ("NN"); for l > 0 the ReLU mask of the layer below rides in the epilogue
                E.gemm_f32(1, dz, w, B, K, n, dz.stride(0), w.stride(0), dprev, K,
                           mask=a_in if l > 0 else None, ldmask=a_in.stride(0) if l > 0 else 0)
                if l > 0:
                    dz = dprev
                else:
                    dx = dprev
        return (dx, None, *grads)


def mlp_forward(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Evaluate an `util.mlp(...)` Sequential (Linear / ReLU / optional final Tanh / optional Squeeze) on the device.
    Differentiable when torch's grad mode is on and a parameter or the input requires a gradient."""
    _need_device(x)
    if x.dtype != torch.float32:
        raise RuntimeError(f"expected float32 input, got {x.dtype}")
    lead = x.shape[:-1]
    h = x.reshape(-1, x.shape[-1])
    h = h if h.stride(-1) == 1 and (h.shape[0] == 1 or h.stride(0) >= h.shape[1]) else h.contiguous()
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    for m in seq:
        if not isinstance(m, (nn.Linear, nn.ReLU, nn.Tanh, Squeeze)):
            raise NotImplementedError(f"{type(m).__name__} inside an mlp() chain is not on the device path")
    if h.shape[1] != lin[0].in_features:
        raise RuntimeError(f"expected (..., {lin[0].in_features}), got {tuple(x.shape)}")
    last = [m for m in seq if not isinstance(m, Squeeze)][-1]
    out_act = 2 if isinstance(last, nn.Tanh) else 1 if isinstance(last, nn.ReLU) else 0
    params = []
    for m in lin:
        if m.weight.device != x.device:
            raise RuntimeError(f"module on {m.weight.device}, input on {x.device}")
        params += [m.weight, m.bias]
    y = _MlpFn.apply(h, out_act, *params).reshape(*lead, lin[-1].out_features)
    if any(isinstance(m, Squeeze) for m in seq):
        y = y.squeeze(-1)
    return y
