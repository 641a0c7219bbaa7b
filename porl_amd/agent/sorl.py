"""SORL agent — drop-in for /root/reference/agent/sorl.py:20-152 (backbone=None) on one MI355X.

Same constructor and attribute names (`v_net`, `policy`, `v_tgt`, `v_optimizer`, `policy_optimizer`,
`lr_schedule`), `update`, `vf_update`, `select_action`.  Value step == POR's; the policy step is
advantage-weighted behaviour cloning of the dataset actions with a tanh-bounded mean and
weight = min(exp(alpha * adv), 100) — alpha MULTIPLIES here (sorl.py:104), unlike POR.

`policy_update` is broken upstream (NameError on `target_v`, sorl.py:163) and is not provided.
"""
from __future__ import annotations

import copy

import torch

from ._iql import ArenaAdam, CosineSchedule, IqlAgentBase
from .policy import BoundedGaussianPolicy
from .value_functions import TwinV

EXP_ADV_MAX = 100.


class SORL(IqlAgentBase):
    def __init__(agent, args, max_steps, tau, alpha, device=torch.device('cpu'), backbone=None,
                 value_lr=1e-4, policy_lr=1e-4, discount=0.99, beta=0.005):
        super().__init__()
        if backbone is not None:
            raise NotImplementedError("SORL(backbone=...) (FasterNet encoder) is outside the accelerated path")
        agent.device = torch.device(device)
        agent.backbone = None
        # SORL builds the value net BEFORE the policy (sorl.py:37-45)
        agent.v_net = TwinV(args.state_size, layer_norm=args.layer_norm,
                            hidden_dim=args.hidden_dim, n_hidden=args.n_hidden)
        agent.policy = BoundedGaussianPolicy(args.state_size, args.action_size,
                                             hidden_dim=args.hidden_dim, n_hidden=args.n_hidden)
        agent.v_tgt = copy.deepcopy(agent.v_net).requires_grad_(False)
        agent._setup_engine(agent.v_net, agent.v_tgt, agent.policy,
                            obs_dim=args.state_size, pol_out_dim=args.action_size, hidden_dim=args.hidden_dim,
                            n_hidden=args.n_hidden, layer_norm=args.layer_norm, pol_tanh=True, weight_mode=1,
                            device=agent.device, max_batch=int(getattr(args, "max_batch", 0) or
                                                               getattr(args, "batch_size", 0) or 1024))
        agent.v_optimizer = ArenaAdam(agent, 0, list(agent.v_net.named_parameters()), value_lr)
        agent.policy_optimizer = ArenaAdam(agent, 1, list(agent.policy.named_parameters()), policy_lr)
        agent.lr_schedule = CosineSchedule(agent.policy_optimizer, max_steps)
        agent.tau = tau
        agent.alpha = alpha
        agent.discount = discount
        agent.beta = beta

    def select_action(agent, observations):
        """Mean action as a numpy array (reference sorl.py:71-76)."""
        agent.flush()
        return agent.policy(observations).mean.cpu().numpy()

    def update(agent, observations, actions, rewards, next_observations, terminals):
        """Joint value + policy step (reference sorl.py:78-128) -> (v_loss, g_loss)."""
        return agent._full_update(observations, next_observations, rewards, terminals, actions,
                                  agent.v_optimizer, agent.policy_optimizer, agent.lr_schedule)

    def vf_update(agent, observations, actions, rewards, next_observations, terminals):
        """Value step only (reference sorl.py:130-152) -> v_loss."""
        agent._value_update(observations, next_observations, rewards, terminals, agent.v_optimizer)
        if agent.async_losses:
            return agent._engine.stats[:1]
        return float(agent._engine.stats[0])

    def update_from_replay(agent, replay, batch_size):
        """Extension (not in the reference): `update` on rows drawn on the device from a PackedReplay."""
        return agent._full_update(None, None, None, None, None, agent.v_optimizer, agent.policy_optimizer,
                                  agent.lr_schedule, replay=replay, batch=batch_size)
