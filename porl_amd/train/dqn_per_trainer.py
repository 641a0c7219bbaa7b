"""Double-DQN with prioritized replay — drop-in for /root/reference/src/porl/train/dqn_per_trainer.py:19-123 on one
MI355X.  `learn()` = sample from the device sum tree, one step kernel (Double-DQN target, weighted TD loss,
backward), Adam, priority write-back with the fresh |TD errors| — nothing returns to the host except the loss.

Quirk reproduced: the reference multiplies importance weights shaped (B, 1) with squared errors shaped (B,), which
broadcasts to (B, B); its loss is therefore mean(is_weights) * mean(td^2) — every sample carries the same weight.
`per_sample_weights=True` selects the evidently intended per-sample form instead.
"""
from __future__ import annotations

import torch

from .. import _native as N
from ..buffer.prioritized_replay_buffer import PrioritizedReplayBuffer
from .cql_trainer import CQLTrainer


class PERTrainer(CQLTrainer):
    def __init__(self, state_size, action_size, gamma, epsilon=1.0, epsilon_min=0.05, epsilon_decay=0.99,
                 update_target_freq=10, device=torch.device("cpu"), log_dir="logs", num_epochs=1000, threshold=0.1,
                 learning_rate=0.0005, batch_size=64, max_batch=4096, capacity=100000, per_sample_weights=False, **kw):
        super().__init__(state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay, update_target_freq, device,
                         log_dir=log_dir, num_epochs=num_epochs, threshold=threshold, alpha=0,
                         learning_rate=learning_rate, batch_size=batch_size, max_batch=max_batch, **kw)
        # dqn_per_trainer.py:62-65
        self.memory = PrioritizedReplayBuffer(capacity, alpha=0.6, beta_start=0.4, beta_frames=100000,
                                              state_shape=(state_size,), device=self.device)
        self.max_initial_priority = 1.0
        self.per_sample_weights = per_sample_weights
        self._td_abs = torch.zeros(max(max_batch, batch_size), dtype=torch.float32, device=self.device)

    def learn(self):
        """dqn_per_trainer.py:67-123 -> loss (float, or the device statistics view with async_losses)."""
        mem, eng = self.memory, self._engine
        mem._flush()
        B = self.batch_size
        # the tree walk / weights / gather of mem.sample, minus the gather: the step kernel reads rows `slots` itself
        states, actions, rewards, next_states, dones, is_w, tree_idx = mem.sample(B)
        del states, actions, rewards, next_states, dones
        slots = (tree_idx - (mem.capacity - 1)).contiguous()
        st = mem._store
        self.optimizer.step_count += 1
        g = self.optimizer.param_groups[0]
        hp = eng.hyper(self.gamma, 0.0, 1.0 / B, self.optimizer.step_count, g["lr"], g["betas"], g["eps"])
        td_abs = self._td_abs[:B]
        if self.per_sample_weights:
            var = N.QnetVariant(1, is_w.data_ptr(), None, td_abs.data_ptr())
        else:
            wmean = is_w.mean().reshape(1)                         # (B,1)*(B,) broadcast of the reference: mean(w)*mean(td^2)
            var = N.QnetVariant(1, None, wmean.data_ptr(), td_abs.data_ptr())
        eng.learn_indexed(hp, st["states"], st["actions"], st["rewards"].view(-1), st["next_states"], st["dones"].view(-1),
                          slots, variant=var)
        mem.update_priorities(tree_idx, td_abs)
        if self.async_losses:
            return eng.stats[:3]
        loss, self.last_td_loss, _ = eng.stats[:3].tolist()
        return loss
