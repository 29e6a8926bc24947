"""numpy-backed tensor: the repo's CPU backend, and the CPU baseline that bench.py times on the GPU box's host cores.

Behaviour follows the reference's CPU tensor (lightgrad/autograd/cpu/tensor.py:4-46) because the golden fixtures
depend on it bit for bit:
  * the payload is an ndarray of exactly the requested dtype (float32 unless told otherwise);
  * `empty / zeros / ones / uniform` build a float64 array first and cast it, so `uniform` consumes numpy's global
    RNG stream the same way for the same seed;
  * `from_numpy` keeps the array's own dtype, `copy` goes back to float32, `numpy()` hands out the live array.
"""
import numpy as np
from ..tensor import AbstractTensor


def _payload(source, dtype) -> np.ndarray:
    """`source` (CpuTensor, ndarray, scalar or nested sequence) as an ndarray whose dtype is exactly `dtype`"""
    want = np.dtype(dtype)
    array = source.data if isinstance(source, CpuTensor) else source
    if isinstance(array, np.ndarray):
        return array if array.dtype == want else array.astype(want)
    return np.asarray(array, dtype=want)


class CpuTensor(AbstractTensor):

    # float32 like the reference.  Tests switch it to float64 to obtain a higher-precision run of the SAME tape (every
    # initialiser, gradient seed and gradient copy then stays in double) - the yardstick fp32 results are measured against
    default_dtype = np.float32

    def __init__(self, data, dtype: type = None, requires_grad: bool = True) -> None:
        AbstractTensor.__init__(self, data=_payload(data, CpuTensor.default_dtype if dtype is None else dtype), requires_grad=requires_grad)

    dtype = property(lambda self: self.data.dtype)
    shape = property(lambda self: self.data.shape)

    @classmethod
    def _from_float64(cls, make, shape, *tensor_args, **tensor_kwargs) -> "CpuTensor":
        # every initialiser: float64 array from numpy, then ONE cast to the tensor's dtype
        return CpuTensor(make(shape), *tensor_args, **tensor_kwargs)

    @staticmethod
    def empty(shape, *tensor_args, **tensor_kwargs) -> "CpuTensor":
        return CpuTensor._from_float64(np.empty, shape, *tensor_args, **tensor_kwargs)

    @staticmethod
    def zeros(shape, *tensor_args, **tensor_kwargs) -> "CpuTensor":
        return CpuTensor._from_float64(np.zeros, shape, *tensor_args, **tensor_kwargs)

    @staticmethod
    def ones(shape, *tensor_args, **tensor_kwargs) -> "CpuTensor":
        return CpuTensor._from_float64(np.ones, shape, *tensor_args, **tensor_kwargs)

    @staticmethod
    def uniform(low, high, shape, *tensor_args, **tensor_kwargs) -> "CpuTensor":
        def draw(size):
            return np.random.uniform(low, high, size=size)
        return CpuTensor._from_float64(draw, shape, *tensor_args, **tensor_kwargs)

    @staticmethod
    def from_numpy(a: np.ndarray, requires_grad: bool = True) -> "CpuTensor":
        a = np.asarray(a)
        return CpuTensor(a, dtype=a.dtype, requires_grad=requires_grad)

    def copy(self, requires_grad: bool = True) -> "CpuTensor":
        # back to the default dtype float32, like the reference's copy (cpu/tensor.py:39-40)
        return CpuTensor(np.array(self.data, copy=True), requires_grad=requires_grad)

    def numpy(self) -> np.ndarray:
        return self.data


# registers all cpu ops (bottom import: ops needs CpuTensor)
from . import ops  # noqa: E402,F401
