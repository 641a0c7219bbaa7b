"""Dueling Double-DQN trainer — drop-in for /root/reference/src/porl/train/dddqn_trainer.py:18-124 on one MI355X:
`DDDQNTrainer(..., network=DuelingQNetwork)` (scripts/train_dddqn_online.py:23), `learn()` = the Double-DQN step of
:59-103 (the online network picks the bootstrap action, the target network values it; plain MSE).  The dueling heads
ride on the Q-network engine as a composed output layer (porl_amd/train/cql_trainer.py:_DuelingHeads)."""
from __future__ import annotations

import torch

from ..net.q_network import DuelingQNetwork
from .dqn_trainer import DDQNTrainer


class DDDQNTrainer(DDQNTrainer):
    def __init__(self, state_size, action_size, gamma, epsilon=1.0, epsilon_min=0.05, epsilon_decay=0.99,
                 update_target_freq=10, device=torch.device("cpu"), network=DuelingQNetwork, log_dir="logs",
                 num_epochs=1000, threshold=0.1, **kw):
        super().__init__(state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay, update_target_freq, device,
                         network=network, log_dir=log_dir, num_epochs=num_epochs, threshold=threshold, **kw)
