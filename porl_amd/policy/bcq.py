"""Discrete BCQ update rules — drop-ins for /root/reference/src/porl/policy/bcq.py:8-86, taking the trainer as their
argument like upstream (`agent.train(env, bcq_learn, dataset=collect_dataset, pretrain=bcq_behavior_pretrain)`).

  bcq_behavior_pretrain : `num_epochs` cross-entropy steps of the behaviour policy on replay minibatches.  The step
      kernel's conservative penalty logsumexp(z) - ln A - z[a] IS the cross-entropy up to the constant ln A, so the
      step runs with the TD term switched off (`td_off`) and alpha = 1 on the behaviour network's own parameter group.
  bcq_learn             : DQN step whose bootstrap action is the target network's best action among those the
      behaviour policy allows in s' (probability > threshold): mask from `BehaviorPolicy.sample`, masked argmax
      inside the step kernel (`next_mask`).
Minibatches come from `agent.replay_buffer.sample` (numpy's index stream, like the reference).
"""
from __future__ import annotations

import math

import torch

from .. import _native as N


def collect_dataset(env, agent, num_episodes: int = 1000) -> None:
    """Random-policy roll-outs into `agent.replay_buffer` (bcq.py:8-20); needs a gymnasium-style environment."""
    for _ in range(num_episodes):
        state, _ = env.reset()
        done = False
        while not done:
            action = env.action_space.sample()
            next_state, reward, done, truncated, _ = env.step(action)
            done = done or truncated
            agent.replay_buffer.push(state, action, reward, next_state, done)
            state = next_state


def _step(agent, eng, opt, batch, alpha, variant):
    states, actions, rewards, next_states, dones = batch
    states, next_states = eng._states(states).contiguous(), eng._states(next_states, "next_states").contiguous()
    B = states.shape[0]
    opt.step_count += 1
    g = opt.param_groups[0]
    hp = eng.hyper(agent.gamma, alpha, 1.0 / B, opt.step_count, g["lr"], g["betas"], g["eps"])
    idx = torch.arange(B, dtype=torch.int64, device=eng.device)
    eng.learn_indexed(hp, states, actions.long().contiguous(), rewards.float().contiguous(), next_states,
                      dones.float().contiguous(), idx, variant=variant)
    return eng.stats


def bcq_behavior_pretrain(agent):
    """bcq.py:23-47 -> list of the per-epoch cross-entropy losses (the reference prints every 10th)."""
    eng = agent._behavior_engine
    ln_a = math.log(agent.action_size)
    var = N.QnetVariant(0, None, None, None, None, 1)                  # td_off: loss = penalty = CE - ln A
    losses = []
    for epoch in range(agent.num_epochs):
        stats = _step(agent, eng, agent.behavior_optimizer, agent.replay_buffer.sample(agent.batch_size), 1.0, var)
        losses.append(stats[2] + ln_a if agent.async_losses else float(stats[2]) + ln_a)
    return losses


def bcq_learn(agent) -> float:
    """bcq.py:50-86 -> loss (float, or the device statistics view with agent.async_losses)."""
    batch = agent.replay_buffer.sample(agent.batch_size)
    mask = agent.behavior_policy.sample(batch[3], agent.threshold)      # (B, A) over next_states
    var = N.QnetVariant(0, None, None, None, mask.data_ptr(), 0)
    stats = _step(agent, agent._engine, agent.optimizer, batch, 0.0, var)
    if agent.async_losses:
        return stats[:3]
    loss, agent.last_td_loss, _ = stats[:3].tolist()
    return loss
