"""DQN and Double-DQN trainers — drop-ins for /root/reference/src/porl/train/dqn_trainer.py:17-118 (`DQNTrainer.learn`,
:93-118) and ddqn_trainer.py:18-99 (`DDQNTrainer.learn`, :58-99) on one MI355X.

Both are the Q-network step of `CQLTrainer` without the conservative penalty (alpha = 0): plain DQN bootstraps from
max_a Q_target(s', a); Double DQN lets the online network choose the action and the target network value it — the
`double_dqn` variant of the one-launch step kernel (csrc/qnet_fused.hpp).  Sampling is the reference's
(`replay_buffer.sample` with numpy's index stream) unless `learn_device_sampled` is used.
"""
from __future__ import annotations

import torch

from .. import _native as N
from .cql_trainer import CQLTrainer


class DQNTrainer(CQLTrainer):
    def __init__(self, state_size, action_size, gamma, epsilon=1.0, epsilon_min=0.05, epsilon_decay=0.99,
                 update_target_freq=10, device=torch.device("cpu"), log_dir="logs", learning_rate=0.0005, replay_buffer=None,
                 batch_size=64, max_batch=4096, **kw):
        super().__init__(state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay, update_target_freq, device,
                         log_dir=log_dir, alpha=0, learning_rate=learning_rate, replay_buffer=replay_buffer,
                         batch_size=batch_size, max_batch=max_batch, **kw)


class DDQNTrainer(DQNTrainer):
    def __init__(self, *args, num_epochs=1000, threshold=0.1, **kw):
        super().__init__(*args, **kw)
        self.num_epochs, self.threshold = num_epochs, threshold

    def learn_on(self, states, actions, rewards, next_states, dones):
        """ddqn_trainer.py:58-99 on an explicit minibatch (device tensors)."""
        eng = self._engine
        states, next_states = eng._states(states), eng._states(next_states, "next_states")
        B = states.shape[0]
        actions = actions.long().contiguous()
        rewards, dones = rewards.float().contiguous(), dones.float().contiguous()
        self.optimizer.step_count += 1
        g = self.optimizer.param_groups[0]
        hp = eng.hyper(self.gamma, 0.0, 1.0 / B, self.optimizer.step_count, g["lr"], g["betas"], g["eps"])
        idx = torch.arange(B, dtype=torch.int64, device=self.device)
        var = N.QnetVariant(1, None, None, None)
        eng.learn_indexed(hp, states.contiguous(), actions, rewards, next_states.contiguous(), dones, idx, variant=var)
        if self.async_losses:
            return eng.stats[:3]
        loss, self.last_td_loss, _ = eng.stats[:3].tolist()
        return loss
