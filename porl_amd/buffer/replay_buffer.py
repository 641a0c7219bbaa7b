"""Replay buffers.

`ReplayBuffer` is the drop-in for /root/reference/buffer/replay_buffer.py:8-79 (== src/porl/buffer/
replaybuffer.py): same constructor, `push`, `sample`, `__len__`, same numpy SoA ring on the host and the
same index stream (`np.random.choice(size, B, replace=False)` under the caller's numpy seed), so runs
are sample-for-sample comparable with the reference.  What changes is WHERE the minibatch is built: the
arrays are mirrored once in HBM and `sample` gathers the B rows on the device (hand-written gather
kernel) instead of fancy-indexing on the host and issuing five H2D copies per step.

`PackedReplay` is the device-resident packed-row store used by POR/SORL (`[s | r | s' | d | a]`, the wire
format of por_train.py:74-78); it can draw its indices on the device (keyed permutation — distinct
indices, O(B)) and shards by rows across data-parallel ranks.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import engine as E
from ..parallel import shard_bounds


class ReplayBuffer:
    def __init__(self, capacity: int, state_shape: tuple, device: torch.device):
        self.capacity = capacity
        self.state_shape = state_shape
        self.device = torch.device(device)
        self.size = 0
        self.position = 0
        self.states = np.zeros((capacity, *state_shape), dtype=np.float32)
        self.actions = np.zeros(capacity, dtype=np.int64)
        self.rewards = np.zeros(capacity, dtype=np.float32)
        self.next_states = np.zeros((capacity, *state_shape), dtype=np.float32)
        self.dones = np.zeros(capacity, dtype=np.float32)
        self._mirror = None          # device copies, rebuilt lazily after pushes
        self._dirty_lo, self._dirty_hi = 0, 0

    def push(self, state, action, reward, next_state, done) -> None:
        p = self.position
        self.states[p] = state
        self.actions[p] = action
        self.rewards[p] = reward
        self.next_states[p] = next_state
        self.dones[p] = float(done)
        if self._mirror is not None:                    # remember which slots the mirror is missing
            self._pending.append(p)
        self.position = (p + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def __len__(self) -> int:
        return self.size

    # -- device mirror ----------------------------------------------------------------------------
    def _flat_states(self, arr):
        return arr.reshape(self.capacity, -1)

    def _sync_mirror(self):
        dev = self.device
        if self._mirror is None:
            self._mirror = dict(
                states=torch.from_numpy(self._flat_states(self.states)).to(dev),
                next_states=torch.from_numpy(self._flat_states(self.next_states)).to(dev),
                # int64 actions travel through the fp32 gather kernel as pairs of 32-bit words
                actions=torch.from_numpy(self.actions).to(dev),
                rewards=torch.from_numpy(self.rewards).to(dev),
                dones=torch.from_numpy(self.dones).to(dev))
            self._pending = []
        elif self._pending:
            idx = np.unique(np.asarray(self._pending, dtype=np.int64))
            tidx = torch.from_numpy(idx).to(dev)
            m = self._mirror
            m["states"][tidx] = torch.from_numpy(self._flat_states(self.states)[idx]).to(dev)
            m["next_states"][tidx] = torch.from_numpy(self._flat_states(self.next_states)[idx]).to(dev)
            m["actions"][tidx] = torch.from_numpy(self.actions[idx]).to(dev)
            m["rewards"][tidx] = torch.from_numpy(self.rewards[idx]).to(dev)
            m["dones"][tidx] = torch.from_numpy(self.dones[idx]).to(dev)
            self._pending = []

    def sample(self, batch_size: int):
        """-> (states, actions, rewards, next_states, dones) on `device`, dtypes f32/i64/f32/f32/f32.
        Raises ValueError when batch_size > len(self) (numpy's behaviour, reference :64)."""
        indices = np.random.choice(self.size, batch_size, replace=False)
        return self.sample_at(indices)

    def sample_at(self, indices):
        if self.device.type != "cuda":
            raise E.N.NativeError("ReplayBuffer.sample gathers on a HIP device (device='cuda'); no CPU path")
        self._sync_mirror()
        idx = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int64)).to(self.device, non_blocking=True)
        return self.gather_device(idx)

    def gather_device(self, idx):
        """The five minibatch tensors for int64 indices that already live on the device."""
        m = self._mirror
        B = idx.numel()
        states = E.gather_rows(m["states"], idx).view(B, *self.state_shape)
        next_states = E.gather_rows(m["next_states"], idx).view(B, *self.state_shape)
        actions = E.gather_rows(m["actions"].view(torch.float32).view(-1, 2), idx).view(torch.int64).view(B)
        rewards = E.gather_rows(m["rewards"].view(-1, 1), idx).view(B)
        dones = E.gather_rows(m["dones"].view(-1, 1), idx).view(B)
        return states, actions, rewards, next_states, dones


class PackedReplay:
    """Device-resident (N, 2S+2+A) fp32 rows; rank r of a data-parallel job keeps rows
    shard_bounds(N, r, world) only (0.5 GB per 1 M rows at S=60; 10 M rows 8-way = 0.62 GB per GPU)."""

    def __init__(self, rows, obs_dim, act_dim, device, rank=0, world=1, seed=0):
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.device = torch.device(device)
        n_total = rows.shape[0]
        self.lo, self.hi = shard_bounds(n_total, rank, world)
        local = rows[self.lo:self.hi]
        self.rows = (torch.from_numpy(np.ascontiguousarray(local)) if isinstance(local, np.ndarray) else local).to(self.device)
        self.n_local, self.width = self.rows.shape
        if self.width != 2 * obs_dim + 2 + act_dim:
            raise ValueError(f"row width {self.width} != 2*{obs_dim}+2+{act_dim}")
        self.seed = (seed * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.draws = 0
        self._idx = None
        self._out = None

    def __len__(self):
        return self.n_local

    def sample_indices(self, batch_size):
        """Distinct local row indices drawn on the device (no host work, no sync)."""
        if batch_size > self.n_local:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        if self._idx is None or self._idx.numel() != batch_size:
            self._idx = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        E.sample_indices(self.n_local, batch_size, self.seed, self.draws, out=self._idx)
        self.draws += 1
        return self._idx

    def gather(self, idx):
        B = idx.numel()
        if self._out is None or self._out.shape[0] != B:
            self._out = torch.empty(B, self.width, dtype=torch.float32, device=self.device)
        return E.gather_rows(self.rows, idx, out=self._out)

    def sample(self, batch_size, indices=None):
        """-> packed (B, row) batch on the device; slice it with `split` like por_train.py:74-78."""
        idx = self.sample_indices(batch_size) if indices is None else indices
        return self.gather(idx)

    def split(self, batch):
        S, A = self.obs_dim, self.act_dim
        return batch[:, :S], batch[:, S], batch[:, S + 1:2 * S + 1], batch[:, 2 * S + 1], batch[:, 2 * S + 2:]
