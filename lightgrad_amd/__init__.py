"""lightgrad_amd - an MI355X-native (gfx950) tensor backend behind lightgrad's
autograd surface.  Top-level names follow the reference's `lightgrad/__init__.py:1-6`."""
from . import autograd, data, loss, nn, optim
from .autograd import Tensor, CpuTensor, HipTensor, Gradients, no_grad

empty, zeros, ones = Tensor.empty, Tensor.zeros, Tensor.ones
uniform, xavier = Tensor.uniform, Tensor.xavier
from_numpy = Tensor.from_numpy
