"""First-class ops of the numpy backend.

Restates the arithmetic of the reference's `lightgrad/autograd/cpu/ops.py`
(op by op; line numbers in each docstring) so that `CpuTensor` here produces
the same numbers as the reference's CPU backend on the same inputs - it is
pinned against fixtures generated from the reference (tests/golden/).

Two gaps of the reference's CPU backend are closed here, following its OpenCL
backend as SURVEY.md §8c prescribes:
  * `sum.backward` (reference cpu/ops.py:293 is a TODO) = broadcast of the
    gradient to the input shape                       (opencl/ops.py:353-368)
  * `dot.backward` for batched operands swaps the last two axes instead of
    reversing all axes                                (opencl/ops.py:127-132)
`conv` (cpu/ops.py:298-356) is restated at the end of this file for the CNN example (SURVEY.md §8f row 4 tail).
"""
import numpy as np
from ..func import Function
from .tensor import CpuTensor


def _raw(x):
    return x.data if isinstance(x, CpuTensor) else x


def _numpy_op(fn_cls):
    """Let an op be written on ndarrays: unwrap tensor arguments, wrap results
    (reference helper `_use_tensor_data`, cpu/ops.py:8-21)."""

    class Op(fn_cls):
        def forward(ctx, *args, **kwargs):
            out = fn_cls.forward(ctx, *[_raw(a) for a in args], **{k: _raw(v) for k, v in kwargs.items()})
            return CpuTensor(data=out, dtype=out.dtype)

        def backward(ctx, out_grad):
            grads = fn_cls.backward(ctx, out_grad.data)
            grads = grads if isinstance(grads, tuple) else (grads,)
            return tuple(CpuTensor(data=g, dtype=g.dtype) for g in grads)

    Op.__name__ = fn_cls.__name__
    Op.__qualname__ = fn_cls.__qualname__
    Op.__doc__ = fn_cls.__doc__
    return Op


def _op(*names, overwrite=False):
    def deco(fn_cls):
        op = _numpy_op(fn_cls)
        for n in (names or (fn_cls.__name__,)):
            CpuTensor.register_op(n, op, overwrite=overwrite)
        return op
    return deco


""" Transformations """


@_op("astype")
class astype(Function):
    """ t.astype(dtype) - not an op of the reference (its tensors change dtype through numpy on the host, cpu/tensor.py:12);
    here so that code written for HipTensor.astype runs on the CPU backend too.  Differentiable between float dtypes. """
    def forward(ctx, a, dtype=np.float32):
        ctx.save_for_backward(a.dtype, np.dtype(dtype))
        return a.astype(dtype)

    def backward(ctx, out_grad):
        src, dst = ctx.get_saved_tensors()
        if src.kind != "f" or dst.kind != "f":
            raise RuntimeError("Cannot Backward through astype(%s -> %s)!" % (src, dst))
        return out_grad.astype(src)


@_op("T", "transpose")
class transpose(Function):
    """ axis permutation view; backward = inverse permutation (cpu/ops.py:25-36) """
    def forward(ctx, a, *axes):
        ctx.save_for_backward(axes)
        return np.transpose(a, axes=(axes if len(axes) > 0 else None))

    def backward(ctx, out_grad):
        axes, = ctx.get_saved_tensors()
        if len(axes) == 0:
            return out_grad.transpose()
        inverse = [0] * len(axes)
        for i, j in enumerate(axes):
            inverse[j] = i
        return out_grad.transpose(*inverse)


@_op()
class reshape(Function):
    """ cpu/ops.py:38-47 """
    def forward(ctx, a, *shape):
        ctx.save_for_backward(a.shape)
        return a.reshape(shape)

    def backward(ctx, out_grad):
        shape, = ctx.get_saved_tensors()
        return out_grad.reshape(shape)


""" Element-wise arithmetic, matrix product, non-linearities

One table instead of one class per op: each row gives the value and the gradients as plain numpy expressions - the
reference's own (cpu/ops.py:52-116, :158-229; operand order kept, so results are its bits) - and says what the backward needs
kept.  `_formula_op` turns a row into a tape node. """

_KEEP_NOTHING, _KEEP_INPUTS, _KEEP_OUTPUT, _KEEP_BOTH = range(4)


def _formula_op(name, value, gradients, keep, aliases=(), overwrite=False, cite=""):
    def forward(ctx, *inputs):
        y = value(*inputs)
        if keep == _KEEP_INPUTS:
            ctx.save_for_backward(*inputs)
        elif keep == _KEEP_OUTPUT:
            ctx.save_for_backward(y)
        elif keep == _KEEP_BOTH:
            ctx.save_for_backward(*(inputs + (y,)))
        return y

    def backward(ctx, out_grad):
        return gradients(out_grad, *ctx.get_saved_tensors())

    node = type(name, (Function,), {"forward": forward, "backward": backward, "__doc__": cite})
    return _op(*((name,) + tuple(aliases)), overwrite=overwrite)(node)


def _last_two_swapped(m):
    return np.swapaxes(m, -1, -2)


def _pow_gradients(g, a, b, y):
    # the exponent's gradient is always formed (NaN for a < 0, as in the reference, cpu/ops.py:101-105)
    with np.errstate(invalid='ignore', divide='ignore'):
        return b * (a ** (b - 1)) * g, g * y * np.log(a)


_FORMULAS = [
    # name        value                                  gradients(g, *kept)                                   kept            extra
    ("neg",       lambda a: -a,                          lambda g: -g,                                          _KEEP_NOTHING,  {}),
    ("add",       lambda a, b: a + b,                    lambda g: (g, g),                                      _KEEP_NOTHING,  {}),
    ("sub",       lambda a, b: a - b,                    lambda g: (g, -g),                                     _KEEP_NOTHING,  {"overwrite": True}),
    ("mul",       lambda a, b: a * b,                    lambda g, a, b: (g * b, a * g),                        _KEEP_INPUTS,   {}),
    ("div",       lambda a, b: a / b,                    lambda g, a, b: (g / b, -a / b**2 * g),                _KEEP_INPUTS,   {"overwrite": True}),
    ("pow",       lambda a, b: a ** b,                   _pow_gradients,                                        _KEEP_BOTH,     {}),
    # a @ b; the backward swaps the last two axes so that it is also right for batched operands (opencl/ops.py:127-132)
    ("dot",       lambda a, b: a @ b,                    lambda g, a, b: (g @ _last_two_swapped(b), _last_two_swapped(a) @ g),
                                                                                                                _KEEP_INPUTS,   {"aliases": ("__matmul__",)}),
    ("sin",       np.sin,                                lambda g, t: np.cos(t) * g,                            _KEEP_INPUTS,   {}),
    ("cos",       np.cos,                                lambda g, t: -np.sin(t) * g,                           _KEEP_INPUTS,   {}),
    ("exp",       np.exp,                                lambda g, y: y * g,                                    _KEEP_OUTPUT,   {}),
    ("log",       np.log,                                lambda g, x: (1 / x) * g,                              _KEEP_INPUTS,   {}),
    ("sigmoid",   lambda t: 1 / (1 + np.exp(-t)),        lambda g, y: y * (1 - y) * g,                          _KEEP_OUTPUT,   {"overwrite": True}),
    ("tanh",      np.tanh,                               lambda g, y: (1 - y**2) * g,                           _KEEP_OUTPUT,   {"overwrite": True}),
    # the gradient passes at exactly 0 (cpu/ops.py:221-229)
    ("relu",      lambda t: np.maximum(t, 0.0),          lambda g, t: g * (t >= 0),                             _KEEP_INPUTS,   {}),
]
for _name, _value, _gradients, _keep, _extra in _FORMULAS:
    globals()[_name] = _formula_op(_name, _value, _gradients, _keep, cite="reference cpu/ops.py, op `%s`" % _name, **_extra)


""" In-place operators and fill: no backward, the result IS the input's storage (cpu/ops.py:120-153) """


def _in_place_op(name, apply, registered_as=None):
    def forward(ctx, t, other):
        apply(t, other)
        return t
    node = type(name, (Function,), {"forward": forward})          # the class name shows in "Cannot Backward through <name>!"
    return _op(registered_as or name, overwrite=True)(node)


iadd = _in_place_op("iadd", lambda t, other: t.__iadd__(other), "__iadd__")
isub = _in_place_op("isub", lambda t, other: t.__isub__(other), "__isub__")
imul = _in_place_op("imul", lambda t, other: t.__imul__(other), "__imul__")
itruediv = _in_place_op("itruediv", lambda t, other: t.__itruediv__(other), "__itruediv__")
fill = _in_place_op("fill", lambda t, value: t.fill(value))


""" Selectors """


def _raw_index(idx):
    if isinstance(idx, tuple):
        return tuple(_raw(i) for i in idx)
    return _raw(idx)


@_op("__getitem__")
class getitem(Function):
    """ cpu/ops.py:234-246 """
    def forward(ctx, a, idx):
        idx = _raw_index(idx)
        ctx.save_for_backward(a.shape, idx)
        return a[idx]

    def backward(ctx, out_grad):
        shape, idx = ctx.get_saved_tensors()
        grad = np.zeros(shape, dtype=CpuTensor.default_dtype)
        parts = idx if isinstance(idx, tuple) else (idx,)
        if any(isinstance(i, (list, range)) or (isinstance(i, np.ndarray) and i.dtype.kind in "iu") for i in parts):
            # integer-array (embedding) index: repeated ids must ACCUMULATE.  The reference's `grad[idx] = out_grad`
            # (cpu/ops.py:245) keeps only the last occurrence; identical whenever ids are unique.
            np.add.at(grad, idx, out_grad)
        else:
            grad[idx] = out_grad
        return grad


@_op("__setitem__")
class setitem(Function):
    """ cpu/ops.py:248-255 """
    def forward(ctx, a, idx, val):
        a[_raw_index(idx)] = val
        return a


""" Reductions """


def _all_axes(x, axis):
    return tuple(range(x.ndim)) if axis is None else axis


@_op()
class max(Function):
    """ every tied maximum receives the full gradient (cpu/ops.py:260-272) """
    def forward(ctx, x, axis=None, keepdims=False):
        axis = _all_axes(x, axis)
        val = np.max(x, axis=axis, keepdims=True)
        ctx.save_for_backward(x, val, axis, keepdims)
        return val if keepdims else np.squeeze(val, axis=axis)

    def backward(ctx, out_grad):
        x, val, axis, keepdims = ctx.get_saved_tensors()
        if not keepdims:
            out_grad = np.expand_dims(out_grad, axis=axis)
        return out_grad * (x == val)


@_op()
class min(Function):
    """ cpu/ops.py:274-286 """
    def forward(ctx, x, axis=None, keepdims=False):
        axis = _all_axes(x, axis)
        val = np.min(x, axis=axis, keepdims=True)
        ctx.save_for_backward(x, val, axis, keepdims)
        return val if keepdims else np.squeeze(val, axis=axis)

    def backward(ctx, out_grad):
        x, val, axis, keepdims = ctx.get_saved_tensors()
        if not keepdims:
            out_grad = np.expand_dims(out_grad, axis=axis)
        return out_grad * (x == val)


@_op()
class sum(Function):
    """ forward = ndarray.sum (cpu/ops.py:288-293); backward = gradient broadcast back to
    the input shape, the semantics of the reference's only sum.backward (opencl/ops.py:353-368) """
    def forward(ctx, t, axis=None, keepdims=False):
        ctx.save_for_backward(t.shape, axis, keepdims)
        return np.asarray(t.sum(axis=axis, keepdims=keepdims))

    def backward(ctx, out_grad):
        shape, axis, keepdims = ctx.get_saved_tensors()
        if not keepdims:
            axes = tuple(range(len(shape))) if axis is None else (axis if isinstance(axis, tuple) else (axis,))
            axes = tuple(sorted(a % len(shape) for a in axes))
            out_grad = np.expand_dims(out_grad, axis=axes) if len(shape) > 0 else out_grad
        return np.broadcast_to(out_grad, shape)


""" Convolution (CNN example; reference cpu/ops.py:298-356) """


def _conv_strides(strides, n):
    """per-window-axis strides; the first window axis is the channel axis and always moves by 1"""
    if isinstance(strides, int):
        return (1,) + (strides,) * (n - 1) if n > 1 else (strides,)
    strides = tuple(strides)
    return (1,) + strides if len(strides) == n - 1 else strides


def _windows(x, kshape, strides):
    """read-only view of all `kshape` windows over the trailing len(kshape) axes: (..., out positions..., window...)"""
    n = len(kshape)
    out = tuple((d - k) // s + 1 for d, k, s in zip(x.shape[-n:], kshape, strides))
    st = x.strides[:-n] + tuple(a * s for a, s in zip(x.strides[-n:], strides)) + x.strides[-n:]
    return np.lib.stride_tricks.as_strided(x, shape=x.shape[:-n] + out + tuple(kshape), strides=st, writeable=False)


@_op()
class conv(Function):
    """ valid N-d cross-correlation: t (..., C, d1..dm), kernel (out_c, C, k1..km) -> (..., out_c, o1..om), computed as
    one matrix product of the window matrix with the flattened kernel; strides apply to the spatial axes """
    def forward(ctx, t, kernel, strides=1):
        n = kernel.ndim - 1
        st = _conv_strides(strides, n)
        assert t.ndim >= n and len(st) == n
        win = _windows(t, kernel.shape[1:], st)
        cols = win.reshape(-1, int(np.prod(kernel.shape[1:])))
        w2 = kernel.reshape(kernel.shape[0], -1)
        y = cols @ w2.T
        ctx.save_for_backward(cols, w2, t.shape, kernel.shape, st, win.shape)
        y = y.reshape(*win.shape[:-n], -1)                     # (..., o0, o1..om, out_c) with o0 the channel position (1)
        return np.ascontiguousarray(np.squeeze(np.swapaxes(y, -n - 1, -1), -1))

    def backward(ctx, out_grad):
        cols, w2, in_shape, k_shape, st, win_shape = ctx.get_saved_tensors()
        n = len(k_shape) - 1
        g2 = np.moveaxis(out_grad, -n, -1).reshape(-1, k_shape[0])
        dw = (g2.T @ cols).reshape(k_shape)
        dwin = (g2 @ w2).reshape(win_shape)
        dx = np.zeros(in_shape, dtype=CpuTensor.default_dtype)
        lead = len(in_shape) - n
        out_pos = win_shape[lead:lead + n]
        # every window offset contributes one strided slab of the input gradient (slabs of one offset never overlap)
        for off in np.ndindex(*k_shape[1:]):
            dst = (Ellipsis,) + tuple(slice(o, o + s * p, s) for o, s, p in zip(off, st, out_pos))
            dx[dst] += dwin[(Ellipsis,) + off]
        return dx, dw
