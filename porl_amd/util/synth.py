"""Synthetic packed-transition data shared by tests, bench.py and the golden generator.

Row wire format (one transition, fp32) is the one the reference's training loop slices
(/root/reference/por_train.py:74-78, sorl_train.py:54-58):

    [ s(S) | r | s'(S) | d | a(A) ]        row = 2*S + 2 + A floats

Distributions follow SURVEY.md §8(d): s ~ N(0,1)^S, s' = s + 0.1*N(0,1)^S, r ~ N(0,1),
d ~ Bernoulli(0.05) stored as 0.0/1.0, a ~ U(-1,1)^A.  numpy's `default_rng` (PCG64) stream is
stable across numpy versions, so a (seed, n_rows, S, A) tuple names the same bytes everywhere.
"""
from __future__ import annotations

import numpy as np


def row_width(obs_dim: int, act_dim: int) -> int:
    return 2 * obs_dim + 2 + act_dim


def make_rows(n_rows: int, obs_dim: int = 60, act_dim: int = 2, seed: int = 0,
              chunk: int = 1 << 18) -> np.ndarray:
    """Return (n_rows, 2*S+2+A) fp32 packed transitions.  Generated in fixed-size chunks so a
    prefix of a longer buffer equals the shorter buffer made with the same seed."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_rows, row_width(obs_dim, act_dim)), dtype=np.float32)
    S, A = obs_dim, act_dim
    for lo in range(0, n_rows, chunk):
        n = min(chunk, n_rows - lo)
        # always draw a full chunk so the stream position does not depend on n_rows
        s = rng.standard_normal((chunk, S), dtype=np.float32)
        e = rng.standard_normal((chunk, S), dtype=np.float32)
        r = rng.standard_normal(chunk, dtype=np.float32)
        d = (rng.random(chunk, dtype=np.float32) < np.float32(0.05)).astype(np.float32)
        a = rng.random((chunk, A), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
        blk = out[lo:lo + n]
        blk[:, :S] = s[:n]
        blk[:, S] = r[:n]
        blk[:, S + 1:2 * S + 1] = s[:n] + np.float32(0.1) * e[:n]
        blk[:, 2 * S + 1] = d[:n]
        blk[:, 2 * S + 2:] = a[:n]
    return out


def split_rows(rows, obs_dim: int, act_dim: int):
    """Slice packed rows the way por_train.py:74-78 does: returns (s, r, s', d, a) views."""
    S, A = obs_dim, act_dim
    return (rows[:, :S], rows[:, S], rows[:, S + 1:2 * S + 1], rows[:, 2 * S + 1], rows[:, 2 * S + 2:])


def make_discrete_transitions(n_rows: int, obs_dim: int = 60, n_actions: int = 10, seed: int = 0):
    """CQL-style SoA transitions (reference ReplayBuffer fields, buffer/replay_buffer.py:26-31):
    states/next_states N(0,1) fp32, actions U{0..A-1} int64, rewards N(0,1), dones Bernoulli(0.05)."""
    rng = np.random.default_rng(seed)
    states = rng.standard_normal((n_rows, obs_dim), dtype=np.float32)
    next_states = rng.standard_normal((n_rows, obs_dim), dtype=np.float32)
    actions = rng.integers(0, n_actions, size=n_rows, dtype=np.int64)
    rewards = rng.standard_normal(n_rows, dtype=np.float32)
    dones = (rng.random(n_rows, dtype=np.float32) < np.float32(0.05)).astype(np.float32)
    return states, actions, rewards, next_states, dones
