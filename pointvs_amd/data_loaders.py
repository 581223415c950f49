"""Batch iteration for the path's callers: per-rank seeded sampling and collation.

The reference draws training samples with `torch.utils.data.WeightedRandomSampler` over
class-balancing weights (/root/reference/point_vs/preprocessing/data_loaders.py:170-186) and
collates with PyG's `DataLoader` (:512-520). It is single-process; under data parallelism
 every rank must see a DIFFERENT part of the SAME draw, reproducibly. Here every
rank draws the identical global sequence from a generator seeded with (seed, epoch) and keeps its
strided share, so the union over ranks is exactly what one process with that seed would have drawn.
Parsing parquet files into graphs stays outside the hot-path scope: a dataset is
any indexable collection of `pointvs_amd.graph.Data`.
"""
import numpy as np
import torch

from .graph import Batch


def class_balance_weights(labels):
    """Per-sample weights 1 / (count of the sample's class), data_loaders.py:173-183. Returns None
    when only one class is present (the reference then uses no sampler)."""
    labels = np.asarray(labels).astype(np.int64)
    active = int(labels.sum())
    if active == 0 or active == len(labels):
        return None
    counts = np.array([len(labels) - active, active], dtype=np.float64)
    return torch.from_numpy((1.0 / counts)[labels])


class RankWeightedSampler:
    """`WeightedRandomSampler(weights, num_samples, replacement=True)` sharded over ranks.

    All ranks draw the same `num_samples` indices (torch.multinomial on a CPU generator seeded with
    seed + epoch) and rank r keeps positions r, r + world, ...; `len()` is the same on every rank
    (the draw is padded by wrapping to a multiple of world), so ranks run the same number of steps -
    a requirement of the gradient all-reduce. weights=None (regression sets, single-class sets): the
    reference then builds its DataLoader with sampler=None and shuffle=False (data_loaders.py:176-178,
    512-520), i.e. iterates in index order every epoch; so does this: the identity sequence, strided
    over the ranks. shuffle=True draws a seeded permutation instead (not a reference behaviour)."""

    def __init__(self, weights, num_samples=None, rank=0, world=1, seed=0, shuffle=False):
        self.weights = None if weights is None else torch.as_tensor(weights, dtype=torch.double)
        if num_samples is None:
            if weights is None:
                raise ValueError('num_samples is required when weights is None')
            num_samples = len(self.weights)
        self.num_samples, self.rank, self.world, self.seed = int(num_samples), int(rank), int(world), int(seed)
        self.epoch = 0
        self.shuffle = bool(shuffle)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def global_draw(self):
        gen = torch.Generator().manual_seed(self.seed + self.epoch)
        if self.weights is None:
            if not self.shuffle:
                return torch.arange(self.num_samples)
            return torch.randperm(self.num_samples, generator=gen)
        return torch.multinomial(self.weights, self.num_samples, True, generator=gen)

    def __len__(self):
        return -(-self.num_samples // self.world)

    def __iter__(self):
        draw = self.global_draw()
        padded = len(self) * self.world
        if padded > draw.numel():
            draw = torch.cat([draw, draw[:padded - draw.numel()]])
        return iter(draw[self.rank::self.world].tolist())


class GraphLoader:
    """Minimal stand-in for the PyG DataLoader the reference uses (data_loaders.py:517-520): batches of
    `batch_size` graphs in sampler (or index) order, collated into one disjoint-union `Batch`;
    drop_last=False like the reference."""

    def __init__(self, dataset, batch_size=32, sampler=None, device=None):
        self.dataset, self.batch_size, self.sampler, self.device = dataset, int(batch_size), sampler, device

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return -(-n // self.batch_size)

    def __iter__(self):
        order = list(self.sampler) if self.sampler is not None else list(range(len(self.dataset)))
        for k in range(0, len(order), self.batch_size):
            batch = Batch.from_data_list([self.dataset[i] for i in order[k:k + self.batch_size]])
            yield batch if self.device is None else batch.to(self.device, non_blocking=True)
