"""MI355X (gfx950) backend: HipTensor + its ops over the liblghip.so C ABI.
The package name `hip` is what makes `AbstractTensor.hip()` appear (tensor.py metaclass)."""
from .tensor import HipTensor, HipDevice, HipBuffer
from .tensor import HipTensor as Tensor
from .lib import HipError
from .graph import HipGraph, GraphedStep
from .profiler import HipProfiler
