"""Backend-independent operators and composite ops.

Restates `lightgrad/autograd/ops.py:10-75` of the reference: python operators
map onto the registered `neg/add/mul/pow` ops, and `sub, div, rsub, rdiv,
sigmoid, tanh, softmax, mean` are WrapperFunctions built from first-class ops,
so a backend that registers only `neg, add, mul, pow, exp, sum, max` gets all of
them (and their gradients) for free.  Backends may overwrite any of them with a
first-class kernel (the CPU backend does for sub/div/sigmoid/tanh,
cpu/ops.py:68, :86, :199, :210).

`pad`/`pool` (reference ops.py:79-148) serve only the CNN example and are out
of the hot-path scope.
"""
from .tensor import AbstractTensor
from .func import WrapperFunction

""" Python operators """

AbstractTensor.__neg__ = lambda t: t.neg()
AbstractTensor.__pow__ = lambda a, b: a.pow(b)
AbstractTensor.__add__ = lambda a, b: a.add(b)
AbstractTensor.__iadd__ = lambda a, b: a.add(b)
AbstractTensor.__radd__ = lambda a, b: a.add(b)
AbstractTensor.__mul__ = lambda a, b: a.mul(b)
AbstractTensor.__imul__ = lambda a, b: a.mul(b)
AbstractTensor.__rmul__ = lambda a, b: a.mul(b)


def _register(*names):
    """Register one WrapperFunction class under several attribute names."""
    def deco(fn):
        op = WrapperFunction.from_function(fn)
        for n in names:
            AbstractTensor.register_op(n, op)
        return op
    return deco


@_register("__isub__", "__sub__", "sub")
def sub(a, b):
    """ a - b through add and neg """
    return a + (-b)


@_register("__itruediv__", "__truediv__", "div")
def div(a, b):
    """ a / b through mul and pow """
    return a * (b ** -1)


@_register("__rsub__")
def rsub(b, a):
    """ a - b where only b is a tensor (python calls b.__rsub__(a)) """
    return b.__class__.sub(a, b)


@_register("__rtruediv__")
def rdiv(b, a):
    """ a / b where only b is a tensor """
    return b.__class__.div(a, b)


""" Non-linear activations """


@_register("sigmoid")
def sigmoid(t):
    return 1 / (1 + t.neg().exp())


@_register("tanh")
def tanh(t):
    return t.sigmoid() * 2 - 1


@_register("softmax")
def softmax(t, axis: int = -1):
    exps = (t - t.max(axis=axis, keepdims=True)).exp()
    return exps / exps.sum(axis=axis, keepdims=True)


""" Reductions """


@_register("mean")
def mean(t, axis: int = None, keepdims: bool = False):
    s = t.sum(axis=axis, keepdims=keepdims)
    return s * (s.numel() / t.numel())
