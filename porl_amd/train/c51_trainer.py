"""C51 trainer — drop-in for /root/reference/src/porl/train/c51_trainer.py:12-232 on one MI355X: same constructor
(`atom_size`, `v_min`, `v_max`, `network_hidden_sizes`), `support`, `delta_z`, `learn()` (:52-174: target-net
expectation argmax, categorical projection onto the support, cross-entropy against the taken action's
log-probabilities)."""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .. import _native as N
from ..net.categorical_q_network import CategoricalQNetwork
from .dist_trainer import DistTrainerBase


class C51Trainer(DistTrainerBase):
    def __init__(self, state_size, action_size, gamma, epsilon=1.0, epsilon_min=0.05, epsilon_decay=0.99,
                 update_target_freq=10, device=torch.device("cpu"), atom_size: int = 51, v_min: float = -10,
                 v_max: float = 10, network_hidden_sizes: List[int] = [128, 128], log_dir: str = "logs",
                 learning_rate: float = 0.0005, batch_size: int = 64, max_batch: int = 4096, replay_buffer=None):
        if atom_size < 2:
            raise ValueError("atom_size must be at least 2.")
        self.atom_size, self.v_min, self.v_max = atom_size, v_min, v_max
        self.delta_z = (v_max - v_min) / (atom_size - 1)
        q = CategoricalQNetwork(state_size, action_size, atom_size, v_min, v_max, hidden_sizes=network_hidden_sizes)
        t = CategoricalQNetwork(state_size, action_size, atom_size, v_min, v_max, hidden_sizes=network_hidden_sizes)
        self._setup(q, t, state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay, update_target_freq, device,
                    learning_rate, log_dir, batch_size, max_batch, replay_buffer)
        self.support = torch.linspace(v_min, v_max, atom_size).to(self.device)

    def learn_on(self, states, actions, rewards, next_states, dones):
        B, actions, rewards, dones = self._load((states, actions, rewards, next_states, dones))
        lc = self._forward_loaded(0, 0, True, self._out[0][:B])
        lt = self._forward_loaded(1, 1, False, self._out[2][:B])
        dl = self._dout[:B]
        N.check(self._engine._lib.porl_c51_loss(N.ptr(lc), N.ptr(lt), lc.stride(0), N.ptr(actions), N.ptr(rewards),
                                                N.ptr(dones), N.ptr(self.support), B, self.action_size, self.atom_size,
                                                self.gamma, float(self.v_min), float(self.v_max), N.ptr(dl),
                                                N.ptr(self._row_loss), N.current_stream_ptr(self.device)), "porl_c51_loss")
        return self._backward_and_step(dl, B)

    def select_action(self, state: np.ndarray) -> int:
        if np.random.rand() < self.epsilon:
            return int(np.random.randint(self.action_size))
        x = torch.from_numpy(np.asarray(state)).float().unsqueeze(0).to(self.device)
        return self._greedy(self.q_network.get_q_values(x))
