"""POR agent — drop-in for /root/reference/agent/por.py:20-112 on one MI355X.

Same constructor, attributes (`goal_policy`, `vf`, `v_target`, `v_optimizer`, `goal_policy_optimizer`,
`goal_lr_schedule`, `tau`, `alpha`, `discount`, `beta`), `state_dict()` keys and
`por_residual_update(observations, next_observations, rewards, terminals) -> (v_loss, g_loss)`.
All arithmetic of the update (and of `goal_policy(obs)`, `vf.both(obs)`) runs in hand-written gfx950 kernels
(porl_amd/csrc).  Torch tensor ops on the device remain only in off-path conveniences: `DiagGaussian.log_prob / sample`
on caller-supplied values (agent/policy.py), the stand-alone EMA on odd-sized tensors (util/util.py), and the replay
mirror's scatter of newly pushed transitions (buffer/replay_buffer.py).

Deliberate differences from the reference (SURVEY.md §8 notes 6):
  * the `pdb.set_trace()` on NLL <= 0 (por.py:104-105) becomes a one-time RuntimeWarning;
  * dead upstream code (`pretrain*`, `por_qlearning_update`, `save/load`) is not provided;
  * with `backbone=FasterNet(...)` (por.py:46-57,75-79) the encoder runs forward-only: it joins no optimizer
    upstream, so its backward only fills gradients nobody reads.
"""
from __future__ import annotations

import copy

import torch

from ._iql import ArenaAdam, CosineSchedule, IqlAgentBase
from .policy import GaussianPolicy
from .value_functions import TwinV

EXP_ADV_MAX = 100.


class POR(IqlAgentBase):
    def __init__(agent, args, max_steps, tau, alpha, backbone=None, device=torch.device('cpu'),
                 value_lr=1e-4, policy_lr=1e-4, discount=0.99, beta=0.005):
        super().__init__()
        agent.device = torch.device(device)
        agent.backbone = None
        # with a backbone the heads see its features but the goal policy still predicts the raw next state (por.py:46-57)
        in_dim = args.state_size if backbone is None else args.feature_dim
        if backbone is not None:
            agent.backbone = backbone.to(agent.device)
        # construction order fixes RNG consumption and state_dict order (por.py:36-45)
        agent.goal_policy = GaussianPolicy(in_dim, args.state_size,
                                           hidden_dim=args.hidden_dim, n_hidden=args.n_hidden)
        agent.vf = TwinV(in_dim, layer_norm=args.layer_norm,
                         hidden_dim=args.hidden_dim, n_hidden=args.n_hidden)
        agent.v_target = copy.deepcopy(agent.vf).requires_grad_(False)
        agent._setup_engine(agent.vf, agent.v_target, agent.goal_policy,
                            obs_dim=in_dim, pol_out_dim=args.state_size, hidden_dim=args.hidden_dim,
                            n_hidden=args.n_hidden, layer_norm=args.layer_norm, pol_tanh=False, weight_mode=0,
                            device=agent.device, max_batch=int(getattr(args, "max_batch", 0) or
                                                               getattr(args, "batch_size", 0) or 1024))
        agent.v_optimizer = ArenaAdam(agent, 0, list(agent.vf.named_parameters()), value_lr)
        agent.goal_policy_optimizer = ArenaAdam(agent, 1, list(agent.goal_policy.named_parameters()), policy_lr)
        agent.goal_lr_schedule = CosineSchedule(agent.goal_policy_optimizer, max_steps)
        agent.tau = tau
        agent.alpha = alpha
        agent.discount = discount
        agent.beta = beta
        agent.step = 0
        agent.pretrain_step = 0

    def por_residual_update(agent, observations, next_observations, rewards, terminals):
        """One POR gradient step (reference por.py:73-112): IQL value step, EMA target update, then the
        advantage-weighted goal-policy regression on s'.  Returns (v_loss, g_loss) as Python floats."""
        target = next_observations
        if agent.backbone is not None:          # por.py:75-79: s then s' are encoded; the regression target stays the raw
            observations = agent.backbone(observations)          # next state (after the encoder's in-place > 8 clamp)
            next_observations = agent.backbone(next_observations)
        return agent._full_update(observations, next_observations, rewards, terminals, target,
                                  agent.v_optimizer, agent.goal_policy_optimizer, agent.goal_lr_schedule)

    def update_from_replay(agent, replay, batch_size):
        """Extension (not in the reference): one POR step on `batch_size` distinct rows drawn on the device
        from a `porl_amd.buffer.replay_buffer.PackedReplay` — sampling, gather and the update without any
        host-side tensor work.  Same arithmetic as `por_residual_update` on those rows."""
        if agent.backbone is not None:
            raise NotImplementedError("update_from_replay draws packed [s | r | s' | d | a] rows for the heads; with a "
                                      "backbone, gather the rows and call por_residual_update")
        return agent._full_update(None, None, None, None, None, agent.v_optimizer, agent.goal_policy_optimizer,
                                  agent.goal_lr_schedule, replay=replay, batch=batch_size)
