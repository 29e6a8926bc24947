"""Autograd host: tape, op registry, backend registry (mirror of the reference's
`lightgrad/autograd/__init__.py:1-9`, with `HipTensor` in the place of
`OpenCLTensor`).  Importing this package never touches the GPU: the HIP
library is loaded on first use of a HipTensor and fails loudly if missing."""
from .grads import Gradients
from .func import Function, WrapperFunction
from .tensor import AbstractTensor
from .cpu import CpuTensor
from .hip import HipTensor, HipDevice, HipGraph

Tensor = CpuTensor      # default backend, as in the reference (autograd/__init__.py:9)
no_grad = Gradients.no_grad
