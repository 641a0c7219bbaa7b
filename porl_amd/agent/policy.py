"""Diagonal-Gaussian policies with the reference's interface (/root/reference/agent/policy.py:12-73).

`forward(obs)` computes the mean on the HIP engine and returns a light distribution object with the
members the reference's callers use (`log_prob`, `mean`, `sample`; por.py:102-103, sorl.py:75,107-108).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..util.util import mlp
from .value_functions import _EngineBacked

LOG_STD_MIN = -5.0
LOG_STD_MAX = 2.0


class DiagGaussian:
    """MultivariateNormal(mean, scale_tril=diag(std)) restricted to what the callers need."""

    def __init__(self, mean, std):
        self.mean, self.stddev = mean, std

    def log_prob(self, value):
        if torch.isnan(value).any():          # MultivariateNormal validate_args behaviour
            raise ValueError("Expected value argument to be within the support of the distribution")
        z = (value - self.mean) / self.stddev
        d = self.mean.shape[-1]
        return -0.5 * (d * math.log(2.0 * math.pi) + (z * z).sum(-1)) - torch.log(self.stddev).sum()

    def sample(self):
        with torch.no_grad():
            return self.mean + self.stddev * torch.randn_like(self.mean)

    rsample = sample


class GaussianPolicy(_EngineBacked):
    _tanh = False

    def __init__(self, obs_dim, act_dim, hidden_dim=256, n_hidden=2):
        super().__init__()
        self.net = mlp([obs_dim, *([hidden_dim] * n_hidden), act_dim],
                       output_activation=nn.Tanh if self._tanh else None)
        self.log_std = nn.Parameter(torch.zeros(act_dim, dtype=torch.float32))
        self._spec = (obs_dim, act_dim, hidden_dim, n_hidden)

    def _attach_private(self, batch):
        from ..engine import IqlEngine
        S, D, H, L = self._spec
        eng = self._private_engine(S, D, H, L, False, self._tanh, batch)
        views = IqlEngine.views(eng.params_pol, eng.tensor_table(IqlEngine.GROUP_POL))
        with torch.no_grad():
            for p, v in zip(self.parameters(), views):      # log_std first, then net.* 
                v.copy_(p)
                p.data = v
        self._engine, self._engine_role, self._private = eng, "policy", True

    def forward(self, obs):
        squeeze = obs.dim() == 1
        if squeeze:
            obs = obs.unsqueeze(0)
        if self._engine is None or (self._private and self._engine.cfg.max_batch < obs.shape[0]):
            self._attach_private(obs.shape[0])
        mean = self._engine.forward_policy(obs)
        if squeeze:
            mean = mean[0]
        std = torch.exp(self.log_std.detach().clamp(LOG_STD_MIN, LOG_STD_MAX))
        return DiagGaussian(mean, std)

    def act(self, obs, deterministic=False, enable_grad=False):
        """policy.py:30-33.  enable_grad=True returns a mean that autograd can differentiate with respect to the
        policy's parameters and `obs` (stand-alone forward through util/hip_mlp.py: same kernels, no engine state)."""
        if enable_grad:
            from ..util.hip_mlp import mlp_forward
            eng = self._engine
            if eng is not None:
                eng.join()                                        # a pipelined update may still be writing the parameters
            with torch.enable_grad():
                mean = mlp_forward(self.net, obs)
                std = torch.exp(self.log_std.clamp(LOG_STD_MIN, LOG_STD_MAX))
            dist = DiagGaussian(mean, std)
            return dist.mean if deterministic else dist.mean + dist.stddev * torch.randn_like(dist.mean)
        dist = self(obs)
        return dist.mean if deterministic else dist.sample()

    def mean_numpy(self, obs):
        """The distribution's mean as a numpy array — what `select_action` returns (reference sorl.py:71-76).  Up to
        8 observations take the host-memory fast path of the engine (three launches, no copies); larger batches the
        ordinary forward."""
        squeeze = obs.ndim == 1
        if squeeze:
            obs = obs[None]
        B = int(obs.shape[0])
        if self._engine is None or (self._private and self._engine.cfg.max_batch < B):
            self._attach_private(B)
        if B <= self._engine.SMALL_BATCH:
            out = self._engine.forward_policy_host(obs)
        else:
            if not isinstance(obs, torch.Tensor):
                obs = torch.as_tensor(obs, dtype=torch.float32)
            out = self._engine.forward_policy(obs.to(self._engine.device)).cpu().numpy()
        return out[0] if squeeze else out


class BoundedGaussianPolicy(GaussianPolicy):
    """Tanh-squashed mean (reference policy.py:35-59).  The reference's |mean|>1 pdb trap is unreachable
    (tanh) and is not reproduced."""
    _tanh = True


class DeterministicPolicy(nn.Module):
    """Tanh-bounded deterministic policy (reference policy.py:62-73; unused by the training scripts): forward on the
    device through util/hip_mlp.py (Linear/ReLU chain, tanh in the last product's epilogue), differentiable."""

    def __init__(self, obs_dim, act_dim, hidden_dim=256, n_hidden=2):
        super().__init__()
        self.net = mlp([obs_dim, *([hidden_dim] * n_hidden), act_dim], output_activation=nn.Tanh)

    def forward(self, obs):
        from ..util.hip_mlp import mlp_forward
        return mlp_forward(self.net, obs)

    def act(self, obs, deterministic=False, enable_grad=False):
        with torch.set_grad_enabled(enable_grad):
            return self(obs)
