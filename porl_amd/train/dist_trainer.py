"""Shared host logic of the distributional trainers (QR-DQN, C51) on the Q-network engine's general path:

    sample (reference index stream) -> load -> online forward on s (activations kept) -> forward passes on s' ->
    loss head kernel (csrc/dist_losses.hpp: dL/d(output) + per-row loss) -> backward -> Adam

Every arithmetic step is a HIP kernel behind the C ABI (porl_qnet_forward_loaded / porl_qr_loss / porl_c51_loss /
porl_qnet_backward / porl_qnet_apply / porl_reduce_mean).  The output layer is wider than the one-launch step kernel's
128 columns (actions x quantiles), so these trainers use the grouped-GEMM path like any wide Q-network.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as N
from ..buffer.replay_buffer import ReplayBuffer
from ..utils.logger import Logger
from .cql_trainer import QnetEngine, _FlatAdam


class DistTrainerBase:
    """Common attributes of the reference's DQNTrainer subclasses: q_network, target_network, optimizer,
    replay_buffer, batch_size, gamma, epsilon*, update_target_freq, device, logger."""

    def _setup(self, q_network, target_network, state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay,
               update_target_freq, device, learning_rate, log_dir, batch_size, max_batch, replay_buffer):
        self.state_size, self.action_size = state_size, action_size
        self.device = torch.device(device)
        self.gamma, self.epsilon, self.epsilon_min, self.epsilon_decay = gamma, epsilon, epsilon_min, epsilon_decay
        self.learning_rate, self.update_target_freq = learning_rate, update_target_freq
        self.q_network, self.target_network = q_network, target_network
        s, n_out, hidden = q_network._spec
        self._engine = eng = QnetEngine(s, n_out, hidden, max(max_batch, batch_size), self.device)
        with torch.no_grad():
            for mod, flat, which in ((q_network, eng.params, 0), (target_network, eng.params_tgt, 1)):
                for p, v in zip(mod.parameters(), eng.views(flat)):
                    v.copy_(p)
                    p.data = v
                mod._engine, mod._which = eng, which
            eng.params_tgt.copy_(eng.params)                           # target.load_state_dict(q.state_dict())
        target_network.eval()
        self.optimizer = _FlatAdam(eng, list(q_network.parameters()), learning_rate)
        self.replay_buffer = replay_buffer if replay_buffer is not None else ReplayBuffer(100000, (state_size,), self.device)
        self.batch_size = batch_size
        self.logger = Logger(log_dir=log_dir)
        self.async_losses = False
        mb = eng.cfg.max_batch
        self._out = [torch.empty(mb, n_out, dtype=torch.float32, device=self.device) for _ in range(3)]
        self._dout = torch.empty(mb, n_out, dtype=torch.float32, device=self.device)
        self._row_loss = torch.empty(mb, dtype=torch.float32, device=self.device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    # -- engine pieces ----------------------------------------------------------------------------------
    def _forward_loaded(self, which_params, which_input, keep, out):
        eng = self._engine
        N.check(eng._lib.porl_qnet_forward_loaded(eng._h, which_params, which_input, int(keep), N.ptr(out), out.stride(0),
                                                  N.current_stream_ptr(eng.device)), "porl_qnet_forward_loaded")
        return out

    def _backward_and_step(self, dout, B):
        eng = self._engine
        N.check(eng._lib.porl_qnet_backward(eng._h, N.ptr(dout), dout.stride(0), N.current_stream_ptr(eng.device)),
                "porl_qnet_backward")
        self.optimizer.step_count += 1
        g = self.optimizer.param_groups[0]
        eng.apply(eng.hyper(self.gamma, 0.0, 1.0 / B, self.optimizer.step_count, g["lr"], g["betas"], g["eps"]))
        N.check(eng._lib.porl_reduce_mean(N.ptr(self._row_loss), B, N.ptr(self._loss), N.current_stream_ptr(eng.device)),
                "porl_reduce_mean")
        if self.async_losses:
            return self._loss
        loss = float(self._loss)
        if loss != loss and not bool(((self._actions >= 0) & (self._actions < self.action_size)).all()):
            # the reference's `gather` raises on such a batch (qr_dqn_trainer.py:147, c51_trainer.py:155); the loss-head
            # kernels make no access through a bad index — the row gets a NaN loss term and no gradient
            raise IndexError("action index out of range in the minibatch (valid: 0..%d)" % (self.action_size - 1))
        return loss

    def _load(self, batch):
        states, actions, rewards, next_states, dones = batch
        B = self._engine.load_batch(states, actions, rewards, next_states, dones)
        self._actions = actions = actions.long().contiguous()
        return B, actions, rewards.float().contiguous(), dones.float().contiguous()

    def learn(self):
        return self.learn_on(*self.replay_buffer.sample(self.batch_size))

    def sync_target(self):
        self._engine.sync_target()

    def train_offline(self, policy=None, num_iterations: int = 10000):
        losses = []
        for step in range(num_iterations):
            losses.append(self.learn() if policy is None else policy())
            if step % self.update_target_freq == 0:
                self.sync_target()
        return losses

    def _greedy(self, q_values):
        return int(q_values.argmax(dim=1).item())
