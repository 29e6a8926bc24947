"""Batching over in-memory tensors (the caller of integer-array indexing).

`Dataset` follows the reference's `lightgrad/data.py:7-32` in behaviour - same constructor, `n`, `shuffle()`,
indexing, iteration and `len()`, and the same consumption of numpy's global RNG (one `np.random.permutation(n)` per
epoch), so a seeded epoch yields the same batches - and is backend-agnostic: with `HipTensor`s the permutation is
applied by a device gather (`t[perm]`, csrc/index.hip) and a batch is a slice view + dense copy, nothing goes through
the host.  The reference's `MNIST` subclass downloads its files (data.py:36-48); there is no network here, so it is
not restated - build a `Dataset` from arrays instead.
"""
from math import ceil
import numpy as np
from .autograd import AbstractTensor


class Dataset(object):

    def __init__(self, tensors, shuffle: bool = True, batchsize: int = 8) -> None:
        self._columns = tuple(tensors)
        if not self._columns or not all(isinstance(t, AbstractTensor) for t in self._columns):
            raise TypeError("Dataset needs at least one tensor")
        rows = {t.shape[0] for t in self._columns}
        assert len(rows) == 1, "all tensors of a Dataset must agree in their first dimension (got %s)" % sorted(rows)
        self._reshuffle, self._batch = bool(shuffle), int(batchsize)

    @property
    def n(self) -> int:
        return self._columns[0].shape[0]

    @property
    def tensors(self) -> tuple:
        return self._columns

    def shuffle(self) -> None:
        order = np.random.permutation(self.n)
        self._columns = tuple(column[order].detach() for column in self._columns)

    def __getitem__(self, idx) -> tuple:
        return tuple(column[idx, ...].detach() for column in self._columns)

    def __len__(self) -> int:
        return ceil(self.n / self._batch)

    def __iter__(self):
        if self._reshuffle:
            self.shuffle()
        for first in range(0, self.n, self._batch):
            yield self[first:first + self._batch]
