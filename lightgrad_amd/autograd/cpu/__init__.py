from .tensor import CpuTensor
from .tensor import CpuTensor as Tensor
