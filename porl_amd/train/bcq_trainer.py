"""Discrete BCQ trainer — drop-in for /root/reference/src/porl/train/bcq_trainer.py:18-82 on one MI355X: a `DQNTrainer`
plus a behaviour-policy model and its optimizer (Adam, lr 5e-4), `num_epochs`, `threshold`.  The two update rules live
in porl_amd/policy/bcq.py like upstream (`bcq_learn`, `bcq_behavior_pretrain`); both run on the one-launch Q-network
step kernel (csrc/qnet_fused.hpp) — the behaviour policy is just another small MLP with its own flat parameter group."""
from __future__ import annotations

import torch

from ..net.behavior_policy import BehaviorPolicy
from ..net.q_network import QNetwork
from .cql_trainer import QnetEngine, _FlatAdam
from .dqn_trainer import DQNTrainer


class BCQTrainer(DQNTrainer):
    def __init__(self, state_size, action_size, gamma, epsilon=1.0, epsilon_min=0.05, epsilon_decay=0.99,
                 update_target_freq=10, device=torch.device("cpu"), network=QNetwork, behavior_policy=BehaviorPolicy,
                 log_dir="logs", num_epochs=1000, threshold=0.1, **kw):
        super().__init__(state_size, action_size, gamma, epsilon, epsilon_min, epsilon_decay, update_target_freq, device,
                         log_dir=log_dir, network=network, **kw)
        self.num_epochs, self.threshold = num_epochs, threshold
        # bcq_trainer.py:59-62 (constructed after the Q-networks: same RNG consumption order)
        self.behavior_policy = behavior_policy(state_size, action_size)
        hidden = self.behavior_policy._spec[2]
        eng = QnetEngine(state_size, action_size, hidden, self._engine.cfg.max_batch, self.device)
        with torch.no_grad():
            for p, v in zip(self.behavior_policy.parameters(), eng.views(eng.params)):
                v.copy_(p)
                p.data = v
        self.behavior_policy._engine = eng
        self._behavior_engine = eng
        self.behavior_optimizer = _FlatAdam(eng, list(self.behavior_policy.parameters()), 0.0005)

    def train(self, env, policy, num_episodes=1000, max_steps=1000, **kwargs):
        """bcq_trainer.py:64-82: optional dataset collection and behaviour pre-training, then the online loop (which
        needs an environment and is outside this package's scope: offline use goes through `train_offline`)."""
        if "dataset" in kwargs:
            kwargs["dataset"](env, self)
        if "pretrain" in kwargs:
            kwargs["pretrain"](self)
        return self.train_offline(policy=policy, num_iterations=num_episodes)
