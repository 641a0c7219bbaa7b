"""TEST / BASELINE INFRASTRUCTURE — not part of the product (only bench.py's `cpu_baseline` leg and tests import this).

Eager PyTorch-CPU restatements of the two update steps, written the way the reference writes them (autograd +
torch.optim.Adam on nn.Sequential MLPs) but from this repository's own description of the math, so that the CPU
baseline on the GPU box's host cores is the strongest honest one: stock eager PyTorch with MKL, like the reference
itself runs, instead of the numpy oracle (which is single-threaded outside its GEMMs).

  * PorTorchCpu.update  — reference agent/por.py:73-112 (value step, Polyak target, advantage-weighted goal policy)
  * CqlTorchCpu.learn   — reference src/porl/train/cql_trainer.py:88-124 (TD + CQL(H) penalty, Adam)

They are checked against the numpy oracle (itself pinned to the reference's goldens) in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn


def _mlp(dims, squeeze=False):
    layers = []
    for i in range(len(dims) - 2):
        layers += [nn.Linear(dims[i], dims[i + 1]), nn.ReLU()]
    layers.append(nn.Linear(dims[-2], dims[-1]))
    return nn.Sequential(*layers)


class PorTorchCpu:
    """POR update on CPU tensors.  `sd`: a POR state_dict (numpy or torch), keys as in agent/por.py."""

    def __init__(self, sd, S, H, L, tau=0.9, alpha=10.0, discount=0.99, beta=0.005, lr=1e-4, max_steps=1000):
        t = lambda k: torch.as_tensor(sd[k]).float().clone()
        dims = [S] + [H] * L
        self.v = [_mlp(dims + [1]), _mlp(dims + [1])]
        self.pol = _mlp(dims + [S])
        self.log_std = nn.Parameter(t("goal_policy.log_std"))
        with torch.no_grad():
            for i, net in enumerate(self.v):
                for j, lin in enumerate(m for m in net if isinstance(m, nn.Linear)):
                    lin.weight.copy_(t(f"vf.v{i + 1}.{2 * j}.weight")); lin.bias.copy_(t(f"vf.v{i + 1}.{2 * j}.bias"))
            for j, lin in enumerate(m for m in self.pol if isinstance(m, nn.Linear)):
                lin.weight.copy_(t(f"goal_policy.net.{2 * j}.weight")); lin.bias.copy_(t(f"goal_policy.net.{2 * j}.bias"))
        self.vt = copy.deepcopy(self.v)
        for net in self.vt:
            net.requires_grad_(False)
        self.v_opt = torch.optim.Adam([p for n in self.v for p in n.parameters()], lr=lr)
        self.p_opt = torch.optim.Adam([self.log_std] + list(self.pol.parameters()), lr=lr)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.p_opt, max_steps)
        self.tau, self.alpha, self.discount, self.beta, self.S = tau, alpha, discount, beta, S

    def update(self, s, sp, r, d):
        with torch.no_grad():
            next_v = torch.minimum(self.vt[0](sp).squeeze(-1), self.vt[1](sp).squeeze(-1))
        target = r + (1.0 - d) * self.discount * next_v
        v_loss = 0.0
        for net in self.v:
            u = target - net(s).squeeze(-1)
            v_loss = v_loss + torch.mean(torch.abs(self.tau - (u < 0).float()) * u ** 2)
        v_loss = v_loss / 2
        self.v_opt.zero_grad(set_to_none=True)
        v_loss.backward()
        self.v_opt.step()
        with torch.no_grad():
            for tn, n in zip(self.vt, self.v):
                for tp, p in zip(tn.parameters(), n.parameters()):
                    tp.mul_(1.0 - self.beta).add_(p, alpha=self.beta)
            v = torch.minimum(self.v[0](s).squeeze(-1), self.v[1](s).squeeze(-1))
            w = torch.clamp_max(torch.exp((target - v) / self.alpha), 100.0)
        mu = self.pol(s)
        sigma = torch.exp(self.log_std.clamp(-5.0, 2.0))
        z = (sp - mu) / sigma
        nll = 0.5 * (self.S * math.log(2 * math.pi) + (z ** 2).sum(-1)) + torch.log(sigma).sum()
        g_loss = torch.mean(w * nll)
        self.p_opt.zero_grad(set_to_none=True)
        g_loss.backward()
        self.p_opt.step()
        self.sched.step()
        return float(v_loss.detach()), float(g_loss.detach())


class CqlTorchCpu:
    """CQL(H) learn step on CPU tensors.  `sd`: QNetwork state_dict (model.{0,2,..}.weight/bias)."""

    def __init__(self, sd, n_actions, gamma=0.99, alpha=1.0, lr=5e-4):
        ws = sorted({int(k.split(".")[1]) for k in sd})
        dims = [torch.as_tensor(sd[f"model.{ws[0]}.weight"]).shape[1]] + [torch.as_tensor(sd[f"model.{i}.weight"]).shape[0] for i in ws]
        self.q = _mlp(dims)
        with torch.no_grad():
            for lin, i in zip((m for m in self.q if isinstance(m, nn.Linear)), ws):
                lin.weight.copy_(torch.as_tensor(sd[f"model.{i}.weight"]).float())
                lin.bias.copy_(torch.as_tensor(sd[f"model.{i}.bias"]).float())
        self.qt = copy.deepcopy(self.q).requires_grad_(False)
        self.opt = torch.optim.Adam(self.q.parameters(), lr=lr)
        self.gamma, self.alpha, self.A = gamma, alpha, n_actions

    def learn(self, s, a, r, sp, d):
        q = self.q(s)
        qa = q.gather(1, a.view(-1, 1)).squeeze(1)
        with torch.no_grad():
            y = r + self.gamma * self.qt(sp).max(1).values * (1.0 - d)
        td = torch.mean((qa - y) ** 2)
        pen = torch.mean(torch.logsumexp(q, dim=1) - math.log(self.A) - qa)
        loss = td + self.alpha * pen
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return float(loss.detach())
