"""SORL agent — drop-in for /root/reference/agent/sorl.py:20-152 on one MI355X, with or without the costmap
encoder as `backbone` (sorl_train.py:29-33: `FasterNet(3, args.feature_dim)`, porl_amd/agent/fasternet.py).

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
        agent.device = torch.device(device)
        agent.backbone = backbone
        # with a backbone the heads see its features (sorl.py:46-56); it joins no optimizer (sorl.py:58-64)
        in_dim = args.state_size if backbone is None else args.feature_dim
        if backbone is not None:
            agent.backbone = backbone.to(agent.device)
        # SORL builds the value net BEFORE the policy (sorl.py:37-45)
        agent.v_net = TwinV(in_dim, layer_norm=args.layer_norm,
                            hidden_dim=args.hidden_dim, n_hidden=args.n_hidden)
        agent.policy = BoundedGaussianPolicy(in_dim, args.action_size,
                                             hidden_dim=args.hidden_dim, n_hidden=args.n_hidden)
        agent.v_tgt = copy.deepcopy(agent.v_net).requires_grad_(False)
        agent._setup_engine(agent.v_net, agent.v_tgt, agent.policy,
                            obs_dim=in_dim, pol_out_dim=args.action_size, hidden_dim=args.hidden_dim,
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
        if agent.backbone is not None:
            observations = agent.backbone(observations)
        return agent.policy.mean_numpy(observations)

    def update(agent, observations, actions, rewards, next_observations, terminals):
        """Joint value + policy step (reference sorl.py:78-128) -> (v_loss, g_loss).  With a backbone both
        observation batches are encoded first, s then s' (sorl.py:81-83); the encoder is forward-only because
        nothing ever consumes its gradients."""
        if agent.backbone is not None:
            observations = agent.backbone(observations)
            next_observations = agent.backbone(next_observations)
        return agent._full_update(observations, next_observations, rewards, terminals, actions,
                                  agent.v_optimizer, agent.policy_optimizer, agent.lr_schedule)

    def vf_update(agent, observations, actions, rewards, next_observations, terminals):
        """Value step only (reference sorl.py:130-152) -> v_loss."""
        if agent.backbone is not None:
            observations = agent.backbone(observations)
            next_observations = agent.backbone(next_observations)
        agent._value_update(observations, next_observations, rewards, terminals, agent.v_optimizer)
        if agent.async_losses:
            return agent._engine.stats[:1]
        return float(agent._engine.stats[0])

    def update_from_replay(agent, replay, batch_size):
        """Extension (not in the reference): `update` on rows drawn on the device from a PackedReplay."""
        return agent._full_update(None, None, None, None, None, agent.v_optimizer, agent.policy_optimizer,
                                  agent.lr_schedule, replay=replay, batch=batch_size)
