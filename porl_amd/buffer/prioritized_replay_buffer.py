"""Prioritized replay with the sum tree on the device — the counterpart of
/root/reference/src/porl/buffer/prioritized_replay_buffer.py:7-108 (+ sum_tree.py:4-77) for typed transitions.

Same constructor, `add(td_error, state, action, reward, next_state, done)`, `sample(batch_size)` ->
(states, actions, rewards, next_states, dones, is_weights, idxs), `update_priorities(idxs, td_errors)`, `len()`,
beta annealing and ring semantics.  The tree is 2*capacity-1 fp64 values in HBM in the reference's heap layout, so
`idxs` are the reference's tree indices.  The host draws the stratified uniforms from Python's `random` exactly as
the reference does (`random.uniform(a, b)` = a + (b - a) * `random.random()`), the device does the tree walk, the
importance weights and the gather; results are device tensors instead of numpy arrays.

Differences: experiences are typed SoA rows (like ReplayBuffer) rather than arbitrary Python objects; ancestors are
recomputed from their children instead of accumulating rounded differences (sums agree to ~1e-16 relative, so a
sampled index can differ from the reference only when s falls within that distance of a boundary); the reference's
"resample from the full range when an empty slot is hit" branch is not needed because empty leaves carry zero
priority.  `add` queues on the host and is flushed in one batch before the next `sample`/`update_priorities`.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from .. import _native as N
from .. import engine as E


class PrioritizedReplayBuffer:
    def __init__(self, capacity, alpha=0.6, beta_start=0.4, beta_frames=100000, epsilon=1e-5, state_shape=None,
                 device="cuda"):
        self.capacity, self.alpha, self.beta_start, self.beta_frames = int(capacity), alpha, beta_start, beta_frames
        self.beta, self.epsilon, self.frame_count = beta_start, epsilon, 0
        self.device = E._norm_device(device)
        if self.device.type != "cuda":
            raise N.NativeError("PrioritizedReplayBuffer keeps its tree on a HIP device (device='cuda'); no CPU path")
        self.state_shape = None if state_shape is None else tuple(state_shape)
        self.tree = torch.zeros(2 * self.capacity - 1, dtype=torch.float64, device=self.device)
        self._stamp = torch.zeros(self.capacity, dtype=torch.int32, device=self.device)
        self._store = None
        self.data_pointer, self.n_entries = 0, 0
        self._pending = []                 # (slot, td_error, state, action, reward, next_state, done)

    # -- storage ---------------------------------------------------------------------------------------
    def _alloc(self, state):
        if self.state_shape is None:
            self.state_shape = tuple(np.shape(state))
        sdim = int(np.prod(self.state_shape)) if self.state_shape else 1
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=self.device)
        self._store = dict(states=z(self.capacity, sdim), next_states=z(self.capacity, sdim), rewards=z(self.capacity, 1),
                           dones=z(self.capacity, 1), actions=z(self.capacity, dt=torch.int64))

    def add(self, td_error, *experience):
        state, action, reward, next_state, done = experience
        if self._store is None:
            self._alloc(state)
        self._pending.append((self.data_pointer, float(td_error), np.asarray(state, dtype=np.float32).reshape(-1),
                              int(action), float(reward), np.asarray(next_state, dtype=np.float32).reshape(-1), float(done)))
        self.data_pointer += 1
        if self.data_pointer >= self.capacity:
            self.data_pointer = 0
        if self.n_entries < self.capacity:
            self.n_entries += 1

    def _flush(self):
        if not self._pending:
            return
        slots = torch.tensor([p[0] for p in self._pending], dtype=torch.int64, device=self.device)
        st = self._store
        # a slot overwritten twice within one flush keeps the later experience (index_copy_ applies in order on one stream)
        last = {}
        for i, p in enumerate(self._pending):
            last[p[0]] = i
        keep = sorted(last.values())
        ks = torch.tensor([self._pending[i][0] for i in keep], dtype=torch.int64, device=self.device)
        st["states"][ks] = torch.from_numpy(np.stack([self._pending[i][2] for i in keep])).to(self.device)
        st["next_states"][ks] = torch.from_numpy(np.stack([self._pending[i][5] for i in keep])).to(self.device)
        st["actions"][ks] = torch.tensor([self._pending[i][3] for i in keep], dtype=torch.int64, device=self.device)
        st["rewards"][ks, 0] = torch.tensor([self._pending[i][4] for i in keep], dtype=torch.float32, device=self.device)
        st["dones"][ks, 0] = torch.tensor([self._pending[i][6] for i in keep], dtype=torch.float32, device=self.device)
        td = torch.tensor([p[1] for p in self._pending], dtype=torch.float64, device=self.device)
        self._update(slots + (self.capacity - 1), td)
        self._pending = []

    def _update(self, tree_idx, td_errors):
        n = tree_idx.numel()
        N.check(N.lib().porl_per_update(N.ptr(self.tree), self.capacity, N.ptr(tree_idx), N.ptr(td_errors), n, self.epsilon,
                                        self.alpha, N.ptr(self._stamp), N.current_stream_ptr(self.device)), "porl_per_update")

    # -- reference API ---------------------------------------------------------------------------------
    def total_priority(self):
        self._flush()
        return float(self.tree[0])

    def sample(self, batch_size):
        self._flush()
        if self.n_entries < 1:
            raise ValueError("cannot sample from an empty buffer")
        self.beta = np.min([1.0, self.beta_start + self.frame_count * (1.0 - self.beta_start) / self.beta_frames])
        self.frame_count += 1
        # random.uniform(a, b) == a + (b - a) * random.random(): one draw per segment, in segment order
        u = torch.tensor([random.random() for _ in range(batch_size)], dtype=torch.float64).to(self.device)
        idxs = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        prio = torch.empty(2 * batch_size, dtype=torch.float64, device=self.device)
        w = torch.empty(batch_size, dtype=torch.float32, device=self.device)
        N.check(N.lib().porl_per_sample(N.ptr(self.tree), self.capacity, N.ptr(u), batch_size, self.n_entries, float(self.beta),
                                        N.ptr(idxs), N.ptr(prio), N.ptr(w), N.current_stream_ptr(self.device)), "porl_per_sample")
        slots = idxs - (self.capacity - 1)
        st = self._store
        states = E.gather_rows(st["states"], slots).view(batch_size, *self.state_shape)
        next_states = E.gather_rows(st["next_states"], slots).view(batch_size, *self.state_shape)
        rewards = E.gather_rows(st["rewards"], slots).view(batch_size)
        dones = E.gather_rows(st["dones"], slots).view(batch_size)
        actions = st["actions"][slots]
        self.last_priorities = prio[:batch_size]
        return states, actions, rewards, next_states, dones, w, idxs

    def update_priorities(self, tree_indices, td_errors):
        self._flush()
        idx = torch.as_tensor(tree_indices, dtype=torch.int64, device=self.device).contiguous()
        td = torch.as_tensor(td_errors, device=self.device).to(torch.float64).reshape(-1).contiguous()
        if idx.numel() != td.numel():
            raise ValueError("tree_indices and td_errors differ in length")
        self._update(idx, td)

    def __len__(self):
        return self.n_entries
