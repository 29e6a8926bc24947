"""Module system and the layers on the hot path.

Restates the reference's `lightgrad/nn.py`: `Module` keeps ordered registries of
parameters and sub-modules filled by attribute assignment (nn.py:14-29),
`parameters/named_parameters` walk own parameters first, then sub-modules
(nn.py:31-45), `map_parameters` rebinds every parameter to `fn(p)` (how a model
is moved to a backend: `model.map_parameters(lambda p: p.hip())`, nn.py:47-55),
`load_parameters` accepts tensors of any backend or ndarrays (nn.py:57-76).
`Linear` is `x @ W.T(1, 0) + b` with `xavier` init (nn.py:90-96); `LayerNorm`
is the composite of nn.py:109-124.  `Conv2d` is CNN-only and out of scope.
"""
import numpy as np
from .autograd import Tensor, AbstractTensor


class Module(object):

    def __init__(self):
        object.__setattr__(self, '_params', {})
        object.__setattr__(self, '_modules', {})

    def forward(self, x):
        raise NotImplementedError()

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def __setattr__(self, name, val):
        if isinstance(val, (AbstractTensor, Module)):
            self.register_param_or_module(name, val)
        object.__setattr__(self, name, val)

    def register_param_or_module(self, name, val):
        if isinstance(val, AbstractTensor):
            self._params[name] = val
        elif isinstance(val, Module):
            self._modules[name] = val
        return val

    def unregister_param_or_module(self, name):
        if name in self._params:
            return self._params.pop(name)
        if name in self._modules:
            return self._modules.pop(name)

    def parameters(self):
        yield from self._params.values()
        for m in self._modules.values():
            yield from m.parameters()

    def named_parameters(self, prefix: str = "", separator: str = "."):
        prefix = (prefix + separator) if len(prefix) > 0 else ""
        for name, p in self._params.items():
            yield (prefix + name, p)
        for name, m in self._modules.items():
            yield from m.named_parameters(prefix=prefix + name, separator=separator)

    def map_parameters(self, fn):
        for key, tensor in list(self._params.items()):
            self.__setattr__(key, fn(tensor))
        for m in self._modules.values():
            m.map_parameters(fn)
        return self

    def load_parameters(self, param_dict, prefix: str = "", separator: str = '.') -> None:
        param_dict = dict(param_dict)
        if len(prefix) > 0:
            prefix += separator
        for key, p in list(self._params.items()):
            assert (prefix + key) in param_dict, "%s not found in param dict!" % (prefix + key)
            new_p = param_dict[prefix + key]
            if not isinstance(new_p, p.__class__):
                new_p = new_p.numpy() if isinstance(new_p, AbstractTensor) else new_p
                assert isinstance(new_p, np.ndarray), "Unexpected parameter type %s!" % new_p.__class__.__name__
                new_p = p.__class__.from_numpy(new_p)
            assert p.shape == new_p.shape, "Shapes do not align! (%s != %s)" % (p.shape, new_p.shape)
            self.__setattr__(key, new_p)
        for key, m in self._modules.items():
            m.load_parameters(param_dict, prefix=prefix + key, separator=separator)


class ModuleList(Module, list):

    def __init__(self, *elements):
        Module.__init__(self)
        list.__init__(self, elements)
        for i, e in enumerate(elements):
            self.register_param_or_module(str(i), e)

    def __setitem__(self, i, e):
        assert i < len(self)
        self.unregister_param_or_module(str(i))
        self.register_param_or_module(str(i), e)
        return list.__setitem__(self, i, e)


class Linear(Module):

    def __init__(self, in_feats: int, out_feats: int, bias: bool = True):
        Module.__init__(self)
        self.weight = Tensor.xavier((out_feats, in_feats))
        self.bias = Tensor.xavier((out_feats,)) if bias else None

    def forward(self, x):
        if hasattr(x, "linear"):
            # optional backend op: the same `x @ W.T(1, 0) + b` as one tape node (HipTensor: bias in the GEMM epilogue)
            return x.linear(self.weight, self.bias) if self.bias is not None else x.linear(self.weight)
        y = x @ self.weight.T(1, 0)
        return (y + self.bias) if self.bias is not None else y


class Conv2d(Module):
    """ valid 2-d convolution with optional zero padding: `x.pad(p).conv(w, strides=s) + b` (reference nn.py:98-107) """

    def __init__(self, in_channels: int, out_channels: int, kernelsize: int = 3, stride: int = 1, pad: int = None, bias: bool = True):
        Module.__init__(self)
        self.w = Tensor.xavier((out_channels, in_channels, kernelsize, kernelsize))
        self.b = Tensor.xavier((1, out_channels, 1, 1)) if bias else None
        self.s, self.p = stride, (kernelsize // 2) if pad is None else pad

    def forward(self, x):
        y = (x.pad(self.p) if self.p > 0 else x).conv(self.w, strides=self.s)
        return (y + self.b) if self.b is not None else y


class LayerNorm(Module):

    def __init__(self, shape: tuple, eps: float = 1e-5):
        Module.__init__(self)
        self.shape = tuple(shape) if isinstance(shape, (tuple, list)) else (shape,)
        self.eps = eps
        self.weight = Tensor.ones(self.shape)
        self.bias = Tensor.zeros(self.shape)

    def forward(self, x):
        assert x.shape[-len(self.shape):] == self.shape, \
            "Shape mismatch in layer norm! (%s <-> %s)" % (x.shape, self.shape)
        if len(self.shape) == 1 and hasattr(x, "layer_norm"):
            # optional backend op: the composite below over the last axis as one kernel (HipTensor)
            return x.layer_norm(self.weight, self.bias, eps=self.eps)
        axes = tuple(range(len(x.shape) - len(self.shape), len(x.shape)))
        D = x - x.mean(axis=axes, keepdims=True)
        V = (D * D).mean(axis=axes, keepdims=True)
        return D / (V + self.eps).pow(1 / 2) * self.weight + self.bias
