"""numpy-backed tensor: the repo's CPU backend (and the CPU baseline that is
timed on the GPU box's host cores).

Restates the reference's `lightgrad/autograd/cpu/tensor.py:4-46`: data is an
ndarray coerced to `dtype` (default float32), initialisers draw in float64 and
cast (so `uniform` consumes the numpy global RNG exactly like the reference),
`from_numpy` keeps the array's dtype, `numpy()` returns the live array.
"""
import numpy as np
from ..tensor import AbstractTensor


class CpuTensor(AbstractTensor):

    def __init__(self, data, dtype: type = np.float32, requires_grad: bool = True) -> None:
        if isinstance(data, CpuTensor):
            data = data.data
        if isinstance(data, np.ndarray):
            if data.dtype != dtype:
                data = data.astype(dtype)
        else:
            data = np.asarray(data, dtype=dtype)
        assert isinstance(data, np.ndarray) and (data.dtype == dtype)
        AbstractTensor.__init__(self, data=data, requires_grad=requires_grad)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def shape(self) -> tuple:
        return self.data.shape

    @staticmethod
    def empty(shape, *args, **kwargs) -> "CpuTensor":
        return CpuTensor(np.empty(shape), *args, **kwargs)

    @staticmethod
    def zeros(shape, *args, **kwargs) -> "CpuTensor":
        return CpuTensor(np.zeros(shape), *args, **kwargs)

    @staticmethod
    def ones(shape, *args, **kwargs) -> "CpuTensor":
        return CpuTensor(np.ones(shape), *args, **kwargs)

    @staticmethod
    def uniform(low, high, shape, *args, **kwargs) -> "CpuTensor":
        return CpuTensor(np.random.uniform(low, high, size=shape), *args, **kwargs)

    @staticmethod
    def from_numpy(a: np.ndarray, requires_grad: bool = True) -> "CpuTensor":
        return CpuTensor(data=a, dtype=a.dtype, requires_grad=requires_grad)

    def copy(self, requires_grad: bool = True) -> "CpuTensor":
        # like the reference (cpu/tensor.py:39-40) the copy is coerced to the default dtype float32
        return CpuTensor(self.data.copy(), requires_grad=requires_grad)

    def numpy(self) -> np.ndarray:
        return self.data


# registers all cpu ops (bottom import: ops needs CpuTensor)
from . import ops  # noqa: E402,F401
