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


""" Padding and pooling (CNN example; SURVEY.md §8f row 4 tail) - built from getitem / setitem / reshape / transpose """

from .func import Function  # noqa: E402


@AbstractTensor.register_op()
class pad(Function):
    """ constant padding of the trailing `len(dims)` axes (reference ops.py:79-98); backward = slice the padding off """
    def forward(ctx, t, padding: int, dims: tuple = (-2, -1), value: float = 0.0):
        n = len(dims)
        before, after = padding if isinstance(padding, tuple) else (padding, padding)
        ctx.save_for_backward(before, after, dims)
        out_shape = t.shape[:-n] + tuple(before + s + after for s in t.shape[-n:])
        out = t.__class__.empty(out_shape, dtype=t.dtype).fill(value).detach()
        inner = tuple(slice(0, s) for s in t.shape[:-n]) + tuple(slice(before, before + s) for s in t.shape[-n:])
        out[inner] = t
        return out

    def backward(ctx, out_grad):
        before, after, dims = ctx.get_saved_tensors()
        idx = [slice(0, d) for d in out_grad.shape]
        for i in dims:
            idx[i] = slice(before, out_grad.shape[i] - after)
        return out_grad[tuple(idx)]


@AbstractTensor.register_op()
class pool(Function):
    """ rearranges non-overlapping `kernel` windows of the trailing axes into a leading axis of size prod(kernel), so
    that a reduction over axis 0 pools (reference ops.py:100-133); input is cropped to a multiple of the kernel """
    def forward(ctx, t, kernel: tuple = (2, 2)):
        n, m = len(kernel), len(t.shape)
        cropped = t.shape[:-n] + tuple((d // k) * k for d, k in zip(t.shape[-n:], kernel))
        x = t[tuple(slice(0, d) for d in cropped)]
        ctx.save_for_backward(kernel, cropped, t.shape)
        split = ()
        for d, k in zip(cropped[-n:], kernel):
            split += (d // k, k)
        x = x.reshape(*cropped[:-n], *split)
        # window axes (the odd ones among the split axes) first, then the leading axes, then the window grid
        perm = tuple(range(m - n + 1, m + n, 2)) + tuple(range(m - n)) + tuple(range(m - n, m + n, 2))
        x = x.transpose(*perm)
        window = 1
        for k in kernel:
            window *= k
        return x.reshape(window, *cropped[:-n], *(d // k for d, k in zip(cropped[-n:], kernel)))

    def backward(ctx, out_grad):
        kernel, cropped, in_shape = ctx.get_saved_tensors()
        n, m = len(kernel), len(cropped)
        g = out_grad.reshape(*kernel, *out_grad.shape[1:])
        inverse = tuple(range(n, m)) + sum(((m + i, i) for i in range(n)), ())
        g = g.transpose(*inverse).reshape(*cropped)
        if tuple(cropped) != tuple(in_shape):
            full = out_grad.__class__.zeros(in_shape)
            full[tuple(slice(0, d) for d in cropped)] = g
            return full
        return g


@_register("max_pool")
def max_pool(t, kernel: tuple = (2, 2)):
    return t.pool(kernel=kernel).max(axis=0, keepdims=False)


@_register("min_pool")
def min_pool(t, kernel: tuple = (2, 2)):
    return t.pool(kernel=kernel).min(axis=0, keepdims=False)


@_register("mean_pool")
def mean_pool(t, kernel: tuple = (2, 2)):
    return t.pool(kernel=kernel).mean(axis=0, keepdims=False)
