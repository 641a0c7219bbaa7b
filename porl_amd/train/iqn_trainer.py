"""IQN trainer — drop-in for /root/reference/src/porl/train/iqn_trainer.py:13-149 on one MI355X.

Upstream's class cannot run as shipped: its constructor builds `IQNNetwork(state_size, action_size, embedding_dim,
num_quantiles_k, dueling_network)` (iqn_trainer.py:58-72) against a class that takes four arguments
(iqn_network.py:10), and `learn` / `select_action` call `get_q_values` (iqn_trainer.py:89,98,109,114), which that class
does not define.  How the step is read here — the only reading under which lines 92-134 type-check against the live
network (forward -> (B, N, A)):
  * `get_q_values(states, taus)` is the network's forward: one value per sampled fraction and action;
  * the network is `IQNNetwork(state_size, action_size, embedding_dim, hidden_size)`; `num_quantiles_k` and
    `dueling_network` are accepted and unused (the live network has neither K output quantiles nor a dueling head),
    `hidden_size` (default 512, the network's own default) is an added keyword, as are `max_norm` (the literal 10.0 of
    line 131) and `replay_buffer`;
  * everything else is lines 92-134 as written: tau' ~ U(0,1) (B, N'), current quantiles of the taken actions; tau'' ~
    U(0,1) (B, N''), Double-DQN action choice on the tau''-mean of the ONLINE net at s', target-net quantiles of that
    action, Bellman targets; pairwise quantile-Huber loss (136-149); clip_grad_norm_(10); Adam.
The golden fixture (`tests/golden/iqn_s9_a5.npz`, oracle/gen_golden.py:gen_iqn) runs upstream's own `learn` (constructor
bypassed, `get_q_values` bound to the live network's forward) on fixed minibatches and fractions.

Arithmetic: Linear layers on the fp32-MFMA GEMM, the rest on csrc/iqn.hpp and dist_losses.hpp (iqn_loss_kernel);
parameters, gradients and Adam moments live in one flat buffer each (clip and Adam are single sweeps).  No CPU path.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .. import _native as N
from .. import engine as E
from ..buffer.replay_buffer import ReplayBuffer
from ..net.iqn_network import IQNNetwork, SelectAction
from ..utils.logger import Logger


class _FlatAdam:
    """torch.optim.Adam's arithmetic (csrc: adam_ema_kernel) over ONE flat buffer holding every parameter of a module;
    the module's nn.Parameters and their .grad become views into the flat tensors."""

    def __init__(self, module, lr, betas=(0.9, 0.999), eps=1e-8, storage_only=False):
        ps = list(module.parameters())
        dev = ps[0].device
        ru4 = lambda k: (k + 3) // 4 * 4                  # every tensor starts on a 16-byte boundary (vector loads in the GEMM)
        n = sum(ru4(p.numel()) for p in ps)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        m = 0 if storage_only else n                      # the target network only needs the flat parameter storage
        self.grad = torch.zeros(m, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(m, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(m, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                if not storage_only:
                    p.grad = self.grad[off:off + k].view_as(p)
                off += ru4(k)                                 # padding stays zero: its gradient is never written
        self.params = ps
        self.param_groups = [dict(params=ps, lr=lr, betas=betas, eps=eps)]
        self.step_count = 0
        self._clip = torch.zeros(2, dtype=torch.float32, device=dev)
        self._ws = torch.zeros(256, dtype=torch.float64, device=dev)

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ over all parameters; returns the (device) total norm."""
        N.check(N.lib().porl_grad_clip(N.ptr(self.grad), self.grad.numel(), float(max_norm), N.ptr(self._clip),
                                       N.ptr(self._ws), N.current_stream_ptr(self.grad)), "porl_grad_clip")
        return self._clip[0]

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        E.adam_ema(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, None, g["lr"], self.step_count, g["betas"][0],
                   g["betas"][1], g["eps"], 0.0)

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])


class IQNTrainer:
    def __init__(self, state_size: int, action_size: int, gamma: float, epsilon: float = 1.0, epsilon_min: float = 0.05,
                 epsilon_decay: float = 0.99, update_target_freq: int = 10, device=torch.device("cpu"),
                 network_hidden_sizes: List[int] = [128, 128], learning_rate: float = 5e-4, buffer_size: int = 100000,
                 batch_size: int = 64, kappa: float = 1.0, embedding_dim: int = 64, num_quantiles_k: int = 8,
                 num_quantiles_n_policy: int = 32, num_quantiles_n_prime_loss: int = 8,
                 num_quantiles_n_double_prime_loss: int = 8, dueling_network: bool = True, log_dir: str = "logs",
                 hidden_size: int = 512, max_norm: float = 10.0, replay_buffer=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.NativeError("porl_amd computes on a HIP device only (device=%s); there is no CPU path" % self.device)
        self.state_size, self.action_size = state_size, action_size
        self.gamma, self.epsilon, self.epsilon_min, self.epsilon_decay = gamma, epsilon, epsilon_min, epsilon_decay
        self.learning_rate, self.update_target_freq = learning_rate, update_target_freq
        self.batch_size, self.kappa, self.embedding_dim = batch_size, kappa, embedding_dim
        self.num_quantiles_k, self.dueling_network = num_quantiles_k, dueling_network          # unused: see the module text
        self.num_quantiles_n_policy = num_quantiles_n_policy
        self.num_quantiles_n_prime_loss = num_quantiles_n_prime_loss
        self.num_quantiles_n_double_prime_loss = num_quantiles_n_double_prime_loss
        self.max_norm = max_norm
        self.q_network = IQNNetwork(state_size, action_size, embedding_dim, hidden_size).to(self.device)
        self.target_network = IQNNetwork(state_size, action_size, embedding_dim, hidden_size).to(self.device)
        self.optimizer = _FlatAdam(self.q_network, learning_rate)
        self._target = _FlatAdam(self.target_network, learning_rate, storage_only=True)
        for p in self.target_network.parameters():
            p.requires_grad_(False)
        self.sync_target()                                                     # target.load_state_dict(q.state_dict())
        self.target_network.eval()
        self.replay_buffer = replay_buffer if replay_buffer is not None else ReplayBuffer(buffer_size, (state_size,), self.device)
        self.logger = Logger(log_dir=log_dir)
        self.async_losses = False
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    def sync_target(self):
        self._target.flat.copy_(self.optimizer.flat)

    # -- acting (iqn_trainer.py:82-91) ---------------------------------------------------------------------------------
    def select_action(self, state: np.ndarray) -> int:
        if np.random.rand() < self.epsilon:
            return int(np.random.randint(self.action_size))
        x = torch.from_numpy(np.asarray(state)).float().unsqueeze(0).to(self.device)
        taus = torch.rand(1, self.num_quantiles_n_policy, device=self.device)
        with torch.no_grad():
            q = self.q_network.get_q_values(x, taus).mean(dim=1)
        return int(q.argmax(dim=1).item())

    # -- learning (iqn_trainer.py:92-134) ------------------------------------------------------------------------------
    def learn(self):
        batch = self.replay_buffer.sample(self.batch_size)
        B = batch[0].shape[0]
        taus_prime = torch.rand(B, self.num_quantiles_n_prime_loss, device=self.device)
        taus_double_prime = torch.rand(B, self.num_quantiles_n_double_prime_loss, device=self.device)
        return self.learn_on(*batch, taus_prime=taus_prime, taus_double_prime=taus_double_prime)

    def learn_on(self, states, actions, rewards, next_states, dones, taus_prime, taus_double_prime):
        dev = self.device
        states = states.to(dev).float().reshape(states.shape[0], -1).contiguous()
        next_states = next_states.to(dev).float().reshape(states.shape[0], -1).contiguous()
        actions = actions.to(dev).long().reshape(-1).contiguous()
        rewards = rewards.to(dev).float().reshape(-1).contiguous()
        dones = dones.to(dev).float().reshape(-1).contiguous()
        taus_prime = taus_prime.to(dev).float().contiguous()
        taus_double_prime = taus_double_prime.to(dev).float().contiguous()
        B, n_cur, n_tgt, A = states.shape[0], taus_prime.shape[1], taus_double_prime.shape[1], self.action_size
        if not (actions.shape[0] == rewards.shape[0] == dones.shape[0] == next_states.shape[0] == B ==
                taus_prime.shape[0] == taus_double_prime.shape[0]):
            raise RuntimeError("minibatch tensors disagree on the batch size")
        lib, st = N.lib(), N.current_stream_ptr(dev)
        current = SelectAction.apply(self.q_network.get_q_values(states, taus_prime), actions)             # (B, N')
        with torch.no_grad():
            z_online = self.q_network.get_q_values(next_states, taus_double_prime).contiguous()            # (B, N'', A)
            z_target = self.target_network.get_q_values(next_states, taus_double_prime).contiguous()
            td = torch.empty(B, n_tgt, dtype=torch.float32, device=dev)
            N.check(lib.porl_iqn_target(N.ptr(z_online), N.ptr(z_target), N.ptr(rewards), N.ptr(dones), self.gamma, B, n_tgt,
                                        A, N.ptr(td), None, st), "porl_iqn_target")
            dcur = torch.empty(B, n_cur, dtype=torch.float32, device=dev)
            row_loss = torch.empty(B, dtype=torch.float32, device=dev)
            N.check(lib.porl_iqn_quantile_huber(N.ptr(current.detach()), N.ptr(td), N.ptr(taus_prime), B, n_cur, n_tgt,
                                                float(self.kappa), N.ptr(dcur), N.ptr(row_loss), st),
                    "porl_iqn_quantile_huber")
        self.optimizer.zero_grad()
        current.backward(dcur)
        self.optimizer.clip_grad_norm_(self.max_norm)
        self.optimizer.step()
        N.check(lib.porl_reduce_mean(N.ptr(row_loss), B, N.ptr(self._loss), st), "porl_reduce_mean")
        if self.async_losses:
            return self._loss
        loss = float(self._loss)
        if loss != loss and not bool(((actions >= 0) & (actions < A)).all()):
            raise IndexError("action index out of range in the minibatch (valid: 0..%d)" % (A - 1))    # upstream's gather raises
        return loss

    def train_offline(self, policy=None, num_iterations: int = 10000):
        losses = []
        for step in range(num_iterations):
            losses.append(self.learn() if policy is None else policy())
            if step % self.update_target_freq == 0:
                self.sync_target()
        return losses
