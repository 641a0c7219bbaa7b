"""Data-parallel exchange for the sharded-replay configuration (BASELINE config 4).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU
tests).  The replay buffer is row-sharded, every rank draws B_local rows of its own shard and scales
its loss gradients by 1/(world * B_local), so a SUM all-reduce of the flat gradient group yields the
gradient of the global-batch mean (SURVEY.md §5.8 parity rule).  The reference has no distributed
code; this is a new capability.
"""
from __future__ import annotations

import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _force_default():
    import os
    return os.environ.get("PORL_DP_FORCE", "0") == "1"


class GradExchange:
    """Collectives of one process group.  `active` decides whether the data-parallel code path runs at all: normally
    world_size > 1.  `force=True` (or PORL_DP_FORCE=1 in the environment) keeps it active in an initialised process
    group of ONE rank, so that a single-GPU box can execute the very calls an 8-GPU job makes — reduce-scatter,
    all-gather, all-reduce on both communicators, from both streams — with RCCL underneath (tests/test_rccl_gpu.py,
    `bench.py --gpus 1` with PORL_BENCH_FORCE_DP=1).  The arithmetic is unchanged: a SUM over one rank."""

    def __init__(self, group=None, force=None):
        self.group = group
        self.force = _force_default() if force is None else bool(force)

    @property
    def active(self):
        d = _dist()
        return bool(d) and (d.get_world_size(self.group) > 1 or self.force)

    @property
    def world_size(self):
        d = _dist()
        return d.get_world_size(self.group) if d else 1

    @property
    def rank(self):
        d = _dist()
        return d.get_rank(self.group) if d else 0

    def allreduce_sum_(self, flat: torch.Tensor):
        if self.active:
            d = _dist()
            d.all_reduce(flat, op=d.ReduceOp.SUM, group=self.group)
        return flat

    def allreduce_sum_async(self, flat: torch.Tensor):
        """Start the SUM all-reduce and return its work handle (None when there is nothing to exchange).  The
        collective runs on the backend's own stream; `handle.wait()` orders the current stream behind it."""
        if self.active:
            d = _dist()
            return d.all_reduce(flat, op=d.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def can_shard(self, flat: torch.Tensor):
        """The flat group splits into equal 16-byte-aligned slices for this world size."""
        return self.active and flat.numel() % (4 * self.world_size) == 0

    def slice_of(self, flat: torch.Tensor):
        w, r = self.world_size, self.rank
        n = flat.numel() // w
        return flat[r * n:(r + 1) * n]

    def reduce_scatter_sum(self, flat: torch.Tensor, out: torch.Tensor):
        """out (numel/world floats) <- this rank's slice of the SUM over ranks of `flat`."""
        d = _dist()
        d.reduce_scatter_tensor(out, flat, op=d.ReduceOp.SUM, group=self.group)
        return out

    def all_gather_(self, flat: torch.Tensor):
        """Every rank contributes its own slice of `flat`; afterwards all of `flat` is current everywhere."""
        d = _dist()
        d.all_gather_into_tensor(flat, self.slice_of(flat), group=self.group)
        return flat

    def allreduce_stats_(self, stats: torch.Tensor):
        """stats[0:2] = (v_loss, g_loss) shares -> SUM; stats[2] = min NLL -> MIN."""
        if self.active:
            d = _dist()
            d.all_reduce(stats[0:2], op=d.ReduceOp.SUM, group=self.group)
            d.all_reduce(stats[2:3], op=d.ReduceOp.MIN, group=self.group)
        return stats


def shard_bounds(n_rows: int, rank: int, world: int):
    """Rows [lo, hi) of the replay store owned by `rank` (contiguous, sizes differ by at most 1)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
