"""Device-resident dataset + shuffled epoch loader — the step in front of the update that the reference does with
`CustomDataset` + `DataLoader(shuffle=True)` (/root/reference/dataloader/dataloader.py:10-55, por_train.py:59-63;
un-runnable as shipped: per-sample file loads, debugger traps).

Wire format (por_train.py:71-78): one fp32 row per transition, `[s (S) | r | s' (S) | d | a (A)]`; on disk the reference
keeps CSV files of 100 such rows (dataloader.py:13: S=365, A=2 -> 734 floats).  Here:

    pack_csv_dir(src_dir, out.npy, row_width)   CSV shards -> ONE packed fp32 row file (host, once)
    DeviceDataset(path_or_array, device)        memory-map the file and stream it to HBM in bounded chunks;
                                                a rank of an N-GPU job keeps only rows shard_bounds(N, rank, world)
    EpochLoader(dataset, batch_size)            `for data in loader:` yields (b, row_width) device tensors; an epoch is
                                                one keyed permutation of the rows walked batch by batch (indices
                                                drawn on the device, rows gathered by one kernel), the last batch
                                                is short — DataLoader(shuffle=True, drop_last=False) semantics

The training loop's slicing (`data[:, :S]`, `data[:, S]`, ...) then hands strided views to the agents unchanged.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import engine as E
from ..parallel import shard_bounds


def pack_rows(rows: np.ndarray, out_path: str) -> str:
    """Write an (N, width) array as the packed fp32 row file (.npy, C order)."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    if rows.ndim != 2:
        raise ValueError("rows must be (N, width)")
    np.save(out_path, rows)
    return out_path if out_path.endswith(".npy") else out_path + ".npy"


def pack_csv_dir(src_dir: str, out_path: str, row_width: int) -> str:
    """Concatenate the reference's CSV shards (np.loadtxt(..., delimiter=',').reshape(-1, row_width),
    dataloader.py:19-20) in sorted file order into one packed fp32 row file."""
    parts = []
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith(".csv"):
            continue
        data = np.loadtxt(os.path.join(src_dir, f), delimiter=",", dtype=np.float64)
        parts.append(data.reshape(-1, row_width).astype(np.float32))
    if not parts:
        raise FileNotFoundError(f"no .csv shards in {src_dir}")
    return pack_rows(np.concatenate(parts, axis=0), out_path)


class DeviceDataset:
    """(N, width) fp32 rows resident in HBM.  `source` is a packed row file (memory-mapped, never loaded whole) or an
    array; with world > 1 only the rank's contiguous row shard is kept."""

    def __init__(self, source, device, rank=0, world=1, chunk_rows=1 << 18):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise E.N.NativeError("DeviceDataset keeps the rows on a HIP device (device='cuda'); there is no CPU path")
        rows = np.load(source, mmap_mode="r", allow_pickle=False) if isinstance(source, (str, os.PathLike)) else source
        if rows.ndim != 2 or rows.dtype != np.float32:
            raise ValueError(f"expected (N, width) float32 rows, got {rows.dtype} {tuple(rows.shape)}")
        self.n_total, self.width = int(rows.shape[0]), int(rows.shape[1])
        self.lo, self.hi = shard_bounds(self.n_total, rank, world)
        n = self.hi - self.lo
        self.rows = torch.empty(n, self.width, dtype=torch.float32, device=self.device)
        for a in range(0, n, chunk_rows):                       # bounded host memory: one chunk at a time
            b = min(n, a + chunk_rows)
            self.rows[a:b].copy_(torch.from_numpy(np.array(rows[self.lo + a:self.lo + b], dtype=np.float32)))   # a writable copy of the chunk
        self.rank, self.world = rank, world

    def __len__(self):
        return self.rows.shape[0]


class EpochLoader:
    """Iterating yields (b, width) device tensors covering every row of the dataset exactly once per epoch in a fresh
    keyed-permutation order (`shuffle=False`: storage order).  len() = number of batches per epoch."""

    def __init__(self, dataset: DeviceDataset, batch_size: int, shuffle=True, drop_last=False, seed=0):
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.seed = dataset, batch_size, shuffle, drop_last, seed
        self.epoch = 0

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds, n, B = self.dataset, len(self.dataset), self.batch_size
        epoch = self.epoch
        self.epoch += 1
        for k in range(len(self)):
            first = k * B
            count = min(B, n - first)
            if self.shuffle:
                idx = E.epoch_indices(n, first, count, self.seed, epoch, device=ds.device)
                yield E.gather_rows(ds.rows, idx)
            else:
                yield ds.rows[first:first + count]
