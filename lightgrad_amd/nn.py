"""Module system and the layers on the hot path (the callers of SURVEY.md §8 row a13).

Public surface = the reference's `lightgrad/nn.py` (same method names, argument meaning and iteration order, so a
training script written against lightgrad runs unchanged); the implementation is this repo's own:

  * a module keeps ONE ordered table of children (`_children`: attribute name -> tensor or Module) that attribute
    assignment maintains; every traversal is built on a single generator, `_leaves()`, which yields
    (owner module, attribute name, qualified name, tensor) with a module's own tensors before those of its
    sub-modules (the order of nn.py:31-45, which optimizers and the data-parallel bucket layout rely on);
  * `parameters`, `named_parameters`, `map_parameters` (nn.py:47-55: `model.map_parameters(lambda p: p.hip())` is how
    a model moves to a backend) and `load_parameters` (nn.py:57-76: tensors of any backend or ndarrays) are thin
    loops over `_leaves()` - no per-method recursion;
  * `Linear` / `LayerNorm` use a backend's fused tape op when the tensor class offers one (`linear`, `layer_norm` on
    HipTensor) and the reference's composite expression (nn.py:96, :117-124) otherwise - identical values.
"""
import numpy as np
from .autograd import Tensor, AbstractTensor


def _is_child(value) -> bool:
    return isinstance(value, (AbstractTensor, Module))


class Module(object):

    def __init__(self):
        object.__setattr__(self, "_children", {})

    # ---- the call protocol ------------------------------------------------------------------------------------------

    def forward(self, *inputs):
        raise NotImplementedError("%s does not implement forward()" % type(self).__name__)

    def __call__(self, *inputs, **options):
        return self.forward(*inputs, **options)

    # ---- bookkeeping ------------------------------------------------------------------------------------------------

    def __setattr__(self, name, value):
        if _is_child(value):
            self.register_param_or_module(name, value)
        object.__setattr__(self, name, value)

    def register_param_or_module(self, name, value):
        """file a tensor / sub-module under `name` (re-assigning a name keeps its position in the table)"""
        if _is_child(value):
            self._children[name] = value
        return value

    def unregister_param_or_module(self, name):
        return self._children.pop(name, None)

    def _own(self, kind):
        return [(n, c) for n, c in self._children.items() if isinstance(c, kind)]

    def _leaves(self, scope: str = "", sep: str = "."):
        """every parameter below this module: (owner, attribute name, qualified name, tensor); own tensors first"""
        for name, tensor in self._own(AbstractTensor):
            yield self, name, scope + name, tensor
        for name, sub in self._own(Module):
            yield from sub._leaves(scope + name + sep, sep)

    # ---- the reference's traversal API ------------------------------------------------------------------------------

    def parameters(self):
        return (leaf[3] for leaf in self._leaves())

    def named_parameters(self, prefix: str = "", separator: str = "."):
        scope = prefix + separator if prefix else ""
        return ((qualified, tensor) for _, _, qualified, tensor in self._leaves(scope, separator))

    def map_parameters(self, fn):
        for owner, name, _, tensor in list(self._leaves()):
            setattr(owner, name, fn(tensor))
        return self

    def load_parameters(self, param_dict, prefix: str = "", separator: str = ".") -> None:
        source = dict(param_dict)
        scope = prefix + separator if prefix else ""
        for owner, name, qualified, current in list(self._leaves(scope, separator)):
            if qualified not in source:
                raise AssertionError("no entry %r among the parameters to load (have: %s)" % (qualified, sorted(source)))
            setattr(owner, name, _as_tensor_like(current, source[qualified], qualified))


def _as_tensor_like(current, incoming, what):
    """`incoming` (tensor of any backend, or ndarray) as a tensor of `current`'s class; shapes must agree"""
    if not isinstance(incoming, type(current)):
        array = incoming.numpy() if isinstance(incoming, AbstractTensor) else incoming
        if not isinstance(array, np.ndarray):
            raise AssertionError("cannot load %r from a %s" % (what, type(incoming).__name__))
        incoming = type(current).from_numpy(array)
    if tuple(incoming.shape) != tuple(current.shape):
        raise AssertionError("%r has shape %s, the value to load has %s" % (what, tuple(current.shape), tuple(incoming.shape)))
    return incoming


class ModuleList(Module, list):
    """a python list whose entries are registered under their index (nn.py:78-88)"""

    def __init__(self, *elements):
        Module.__init__(self)
        list.__init__(self, elements)
        for index, element in enumerate(elements):
            self.register_param_or_module(str(index), element)

    def __setitem__(self, index, element):
        if not -len(self) <= index < len(self):
            raise IndexError("ModuleList assignment index out of range")
        index %= len(self)
        # drop, then file again: like the reference (nn.py:84-88) a replaced entry moves to the END of the traversal
        # order, which the per-parameter step count of Adam / AdaBelief (optim.py:36, :48) makes observable
        self.unregister_param_or_module(str(index))
        self.register_param_or_module(str(index), element)
        list.__setitem__(self, index, element)


class Linear(Module):
    """y = x @ W^T + b with W of shape (out, in), both drawn by `xavier` (nn.py:90-96)"""

    def __init__(self, in_feats: int, out_feats: int, bias: bool = True):
        Module.__init__(self)
        self.weight = Tensor.xavier((out_feats, in_feats))
        self.bias = Tensor.xavier((out_feats,)) if bias else None

    def forward(self, x, residual=None):
        """`residual` (extension of the reference's forward(x)): a tensor of the output's shape added to the result, as in
        `dense(h) + h_in` of a transformer block - a backend with a fused op adds it where the product is made"""
        fused = getattr(x, "linear", None)
        if fused is not None:        # one tape node, bias (and residual) added in the GEMM epilogue (HipTensor)
            if residual is not None:
                return fused(self.weight, self.bias, residual)
            return fused(self.weight) if self.bias is None else fused(self.weight, self.bias)
        y = x @ self.weight.T(1, 0)
        y = y if self.bias is None else y + self.bias
        return y if residual is None else y + residual


class Conv2d(Module):
    """valid 2-d convolution, zero padding of kernelsize // 2 unless told otherwise (nn.py:98-107; CNN tail of §8f)"""

    def __init__(self, in_channels: int, out_channels: int, kernelsize: int = 3, stride: int = 1, pad: int = None, bias: bool = True):
        Module.__init__(self)
        self.w = Tensor.xavier((out_channels, in_channels, kernelsize, kernelsize))
        self.b = Tensor.xavier((1, out_channels, 1, 1)) if bias else None
        self.s = stride
        self.p = kernelsize // 2 if pad is None else pad

    def forward(self, x):
        padded = x.pad(self.p) if self.p > 0 else x
        y = padded.conv(self.w, strides=self.s)
        return y if self.b is None else y + self.b


class LayerNorm(Module):
    """normalise over the trailing `shape` axes, then scale and shift (nn.py:109-124)"""

    def __init__(self, shape: tuple, eps: float = 1e-5):
        Module.__init__(self)
        self.shape = tuple(shape) if isinstance(shape, (tuple, list)) else (shape,)
        self.eps = eps
        self.weight = Tensor.ones(self.shape)
        self.bias = Tensor.zeros(self.shape)

    def forward(self, x):
        k = len(self.shape)
        if tuple(x.shape[-k:]) != self.shape:
            raise AssertionError("LayerNorm over %s applied to an input of shape %s" % (self.shape, tuple(x.shape)))
        fused = getattr(x, "layer_norm", None)
        if k == 1 and fused is not None:          # the composite below as one kernel (HipTensor)
            return fused(self.weight, self.bias, eps=self.eps)
        axes = tuple(range(len(x.shape) - k, len(x.shape)))
        centred = x - x.mean(axis=axes, keepdims=True)
        variance = (centred * centred).mean(axis=axes, keepdims=True)
        return centred / (variance + self.eps).pow(1 / 2) * self.weight + self.bias
