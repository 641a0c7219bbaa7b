"""Device-resident dataset + shuffled epoch loader (what por_train.py:59-78 does with
CustomDataset + DataLoader(shuffle=True), drop_last=False)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from porl_amd.util.synth import make_rows

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def test_an_epoch_is_one_permutation_walked_in_batches(tmp_path):
    from porl_amd.dataloader import DeviceDataset, EpochLoader, pack_rows
    S, A, N, B = 60, 2, 1000 + 37, 128
    rows = make_rows(N, S, A, seed=4)
    rows[:, 0] = np.arange(N)                                   # a row id that survives the shuffle
    path = pack_rows(rows, str(tmp_path / "rows.npy"))
    ds = DeviceDataset(path, DEV, chunk_rows=300)               # streamed in 4 chunks
    assert len(ds) == N and ds.width == 2 * S + 2 + A
    assert torch.equal(ds.rows.cpu(), torch.from_numpy(rows))
    loader = EpochLoader(ds, B, seed=3)
    assert len(loader) == (N + B - 1) // B
    orders = []
    for epoch in range(2):
        batches = [b.cpu().numpy() for b in loader]
        assert [len(b) for b in batches] == [B] * (N // B) + [N % B]      # drop_last=False: short last batch
        got = np.concatenate(batches)
        ids = got[:, 0].astype(np.int64)
        assert np.array_equal(np.sort(ids), np.arange(N))                 # every row exactly once
        assert np.array_equal(got, rows[ids])                             # rows travel whole
        orders.append(ids)
    assert not np.array_equal(orders[0], orders[1])                       # a fresh permutation per epoch
    again = EpochLoader(ds, B, seed=3)
    assert np.array_equal(np.concatenate([b.cpu().numpy() for b in again])[:, 0].astype(np.int64), orders[0])
    assert len(EpochLoader(ds, B, drop_last=True)) == N // B
    plain = np.concatenate([b.cpu().numpy() for b in EpochLoader(ds, B, shuffle=False)])
    assert np.array_equal(plain, rows)


def test_ranks_keep_disjoint_shards():
    from porl_amd.dataloader import DeviceDataset
    rows = make_rows(1001, 8, 2, seed=1)
    parts = [DeviceDataset(rows, DEV, rank=r, world=4) for r in range(4)]
    assert sum(len(p) for p in parts) == 1001
    assert np.array_equal(np.concatenate([p.rows.cpu().numpy() for p in parts]), rows)


def test_training_loop_shape_of_the_reference():
    """por_train.py:71-82: slice each loader batch into strided views and update; equals updating on the same rows
    gathered by hand."""
    from porl_amd import engine as E
    from porl_amd.agent.por import POR
    from porl_amd.dataloader import DeviceDataset, EpochLoader
    S, A, N, B = 60, 2, 300, 128
    rows = make_rows(N, S, A, seed=9)
    ds = DeviceDataset(rows, DEV)
    args = SimpleNamespace(state_size=S, hidden_dim=64, n_hidden=2, layer_norm=False, action_size=A, max_batch=B)
    agents = []
    for _ in range(2):
        torch.manual_seed(0)
        agents.append(POR(args, 100, 0.9, 10.0, device=DEV))
    losses = []
    for data in EpochLoader(ds, B, seed=11):
        losses.append(agents[0].por_residual_update(data[:, :S], data[:, S + 1:-A - 1], data[:, S], data[:, -A - 1]))
    drows = torch.from_numpy(rows).to(DEV)
    for k, first in enumerate(range(0, N, B)):
        idx = E.epoch_indices(N, first, min(B, N - first), 11, 0, device=DEV)
        d = drows[idx]
        assert agents[1].por_residual_update(d[:, :S], d[:, S + 1:-A - 1], d[:, S], d[:, -A - 1]) == losses[k]
    assert len(losses) == 3


def test_errors():
    from porl_amd import _native as N
    from porl_amd import engine as E
    from porl_amd.dataloader import DeviceDataset
    with pytest.raises(N.NativeError):
        DeviceDataset(make_rows(10, 4, 1, seed=0), "cpu")
    with pytest.raises(ValueError):
        DeviceDataset(np.zeros((4, 4)), DEV)                    # float64
    with pytest.raises(N.NativeError):
        E.epoch_indices(100, 90, 20, 0, 0, device=DEV)          # walks past the end of the epoch


@pytest.mark.parametrize("n", [2 ** 20 + 1, 2 ** 16 + 1, 4 ** 7, 1])
def test_epoch_permutation_is_exact_just_above_a_power_of_four(n):
    """The keyed permutation is a Feistel bijection on 4^hb >= n points, cycle-walked into [0, n): just above a power of
    four the domain is almost 4x the range and many positions need several rounds (the round-1 version stopped after 64
    rounds and fell back to the identity, which could duplicate a row — ADVICE r1).  Every position of an epoch must map
    to a distinct row, for several keys."""
    import torch
    from porl_amd import engine as E
    dev = torch.device("cuda")
    for seed, epoch in ((0, 0), (123456789, 7), (2 ** 63 - 1, 2 ** 31 + 5)):
        idx = E.epoch_indices(n, 0, n, seed, epoch, device=dev)
        assert idx.dtype == torch.int64 and idx.numel() == n
        assert int(idx.min()) == 0 and int(idx.max()) == n - 1
        assert torch.unique(idx).numel() == n
    if n > 1:                                               # a window in the middle of an epoch equals that slice
        whole = E.epoch_indices(n, 0, n, 5, 3, device=dev)
        part = E.epoch_indices(n, n // 3, min(1000, n - n // 3), 5, 3, device=dev)
        assert torch.equal(part, whole[n // 3:n // 3 + part.numel()])
