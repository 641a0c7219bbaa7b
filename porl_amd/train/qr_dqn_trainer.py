"""QR-DQN trainer — drop-in for /root/reference/src/porl/train/qr_dqn_trainer.py:13-264 on one MI355X: same constructor
(`network_hidden_sizes`, `num_quantiles`, `kappa`, `learning_rate`), `q_network` / `target_network` (QRNetwork),
`tau`, `learn()` (:97-215: Double-DQN action selection by the online net's quantile means, target-net quantiles,
quantile-Huber loss with the reference's tau broadcast) and `select_action` (:217-258)."""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .. import _native as N
from ..net.qr_dqn_network import QRNetwork
from .dist_trainer import DistTrainerBase


class QRDQNTrainer(DistTrainerBase):
    def __init__(self, state_size: int, action_size: int, gamma: float, epsilon: float = 1.0, epsilon_min: float = 0.05,
                 epsilon_decay: float = 0.99, update_target_freq: int = 10, device=torch.device("cpu"),
                 network_hidden_sizes: List[int] = [128, 128], num_quantiles: int = 51, kappa: float = 1.0,
                 learning_rate: float = 5e-4, log_dir: str = "logs", batch_size: int = 64, max_batch: int = 4096,
                 replay_buffer=None):
        self.num_quantiles, self.kappa = num_quantiles, kappa
        q = QRNetwork(state_size, action_size, num_quantiles, network_hidden_sizes)
        t = QRNetwork(state_size, action_size, num_quantiles, network_hidden_sizes)
        self._setup(q, t, state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay, update_target_freq, device,
                    learning_rate, log_dir, batch_size, max_batch, replay_buffer)
        i = torch.arange(0, num_quantiles, device=self.device, dtype=torch.float32)
        self.tau = ((2 * i + 1) / (2 * num_quantiles)).unsqueeze(0)          # (1, N), qr_dqn_trainer.py:89-95

    def learn_on(self, states, actions, rewards, next_states, dones):
        B, actions, rewards, dones = self._load((states, actions, rewards, next_states, dones))
        z_cur = self._forward_loaded(0, 0, True, self._out[0][:B])
        z_no = self._forward_loaded(0, 1, False, self._out[1][:B])
        z_nt = self._forward_loaded(1, 1, False, self._out[2][:B])
        dz = self._dout[:B]
        N.check(self._engine._lib.porl_qr_loss(N.ptr(z_cur), N.ptr(z_no), N.ptr(z_nt), z_cur.stride(0), N.ptr(actions),
                                               N.ptr(rewards), N.ptr(dones), B, self.action_size, self.num_quantiles,
                                               self.gamma, self.kappa, N.ptr(dz), N.ptr(self._row_loss),
                                               N.current_stream_ptr(self.device)), "porl_qr_loss")
        return self._backward_and_step(dz, B)

    def select_action(self, state: np.ndarray) -> int:
        if np.random.rand() < self.epsilon:
            return int(np.random.randint(self.action_size))
        x = torch.from_numpy(np.asarray(state)).float().unsqueeze(0).to(self.device)
        return self._greedy(self.q_network.get_mean_q_values(x))
