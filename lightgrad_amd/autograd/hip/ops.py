"""Ops of the HipTensor backend: each `Function` is a thin host wrapper that
computes shapes/strides and enqueues hand-written gfx950 kernels through the
C ABI (include/lghip.h).  Registered names and forward/backward semantics follow
the reference's backends op by op (cpu/ops.py = numeric ground truth,
opencl/ops.py = view semantics); citations per class.

Nothing here synchronises with the device and nothing falls back to numpy.
"""
import builtins as _py      # this module defines ops named max / min / sum / pow
import ctypes
import os
import numpy as np
from ..func import Function
import weakref
from .tensor import HipTensor, HipBuffer, GradGroup, HeldPair, contiguous_strides, flush_lazy_readers
from . import lib as _l
from .lib import i64

_F32 = np.dtype(np.float32)
_NULL = None


def _is_scalar(x):
    return isinstance(x, (int, float, np.integer, np.floating)) or (isinstance(x, np.ndarray) and x.ndim == 0)


def _require_f32(*tensors):
    for t in tensors:
        if isinstance(t, HipTensor) and t.dtype != _F32:
            raise TypeError("HipTensor arithmetic is float32-only (got %s); layout ops accept any dtype" % t.dtype)


def _broadcast_shapes(*shapes):
    nd = _py.max(len(s) for s in shapes)
    out = [1] * nd
    for s in shapes:
        for i, d in enumerate(s):
            j = nd - len(s) + i
            if d != 1:
                if out[j] != 1 and out[j] != d:
                    raise ValueError("operands could not be broadcast together with shapes %s" % (shapes,))
                out[j] = d
    return tuple(out)


def _bstrides(t, shape):
    """strides of `t` viewed as `shape` (numpy broadcasting: missing / size-1 dims get stride 0)"""
    if t._shape == shape:
        return t._strides
    lead = len(shape) - len(t._shape)
    return (0,) * lead + tuple(0 if s == 1 and o != 1 else st for s, st, o in zip(t._shape, t._strides, shape[lead:]))


def _ew(op, shape, ins, scalar=0.0, n_out=1, out=None):
    """enqueue one lg_ew call; `ins` holds HipTensors or None (= the scalar operand)"""
    L = _l.lib()
    if out is not None:
        flush_lazy_readers(out)          # writing into existing storage: lazy tensors defined from it are computed first
    outs = [out] if out is not None else [HipTensor.empty(shape) for _ in range(n_out)]
    args = []
    for k in range(4):
        t = ins[k] if k < len(ins) else None
        if t is None:
            args += [_NULL, _NULL]
        else:
            args += [t.ptr, i64(_bstrides(t, shape))]
    o1 = outs[1] if len(outs) > 1 else None
    _l.check(L.lg_ew(op, len(shape), i64(shape), outs[0].ptr, i64(outs[0]._strides),
                     o1.ptr if o1 is not None else _NULL, i64(o1._strides) if o1 is not None else _NULL,
                     *args, float(scalar)))
    return outs[0] if len(outs) == 1 else tuple(outs)


# ---- tensors that are not float32 (csrc/typed.hip) --------------------------------------------------------------------
# The reference's tensors keep their numpy dtype (cpu/tensor.py:45-46; labels int16, ids int32) and numpy computes with
# whatever it is given (cpu/ops.py:52-84).  float32 has the tuned kernels; int16 / int32 / int64 / float64 go through
# lg_ew_typed / lg_reduce_typed with numpy's semantics (wrap-around integers, int64 sums) for neg, add, sub, mul (float64: also
# div, pow), their in-place forms, sum / max / min, and `astype`.  Two tensor operands must share a dtype - numpy would
# promote, here a cast is asked for explicitly (`.astype`).
_TYPED = {np.dtype(np.int16): _l.DT_I16, np.dtype(np.int32): _l.DT_I32, np.dtype(np.int64): _l.DT_I64, np.dtype(np.float64): _l.DT_F64}
_CASTABLE = dict(_TYPED)
_CASTABLE[_F32] = _l.DT_F32
_TYPED_OPS_INT = (_l.EW_COPY, _l.EW_NEG, _l.EW_ADD, _l.EW_SUB, _l.EW_MUL)
_TYPED_OPS_F64 = _TYPED_OPS_INT + (_l.EW_DIV, _l.EW_POW)
_F64 = np.dtype(np.float64)


def _operand_dtype(*operands):
    """the dtype shared by the tensor operands (scalars do not count); TypeError when they differ"""
    dt = None
    for t in operands:
        if isinstance(t, HipTensor):
            if dt is not None and t._dtype != dt:
                raise TypeError("HipTensor operands of different dtypes (%s, %s): cast one of them with .astype() first" % (dt, t._dtype))
            dt = t._dtype
    return dt


def _ew_typed(op, dt, shape, a, b, out=None):
    """a (op) b for int16 / int32 / int64 / float64 tensors; a or b may be a python / numpy scalar (None for unary ops)"""
    code = _TYPED.get(dt)
    if code is None or op not in (_TYPED_OPS_F64 if dt == _F64 else _TYPED_OPS_INT):
        raise TypeError("HipTensor: this operation is not defined for dtype %s (float32 everywhere; int16 / int32 / int64 / float64: "
                        "neg, add, sub, mul, sum, max, min, astype, and for float64 div and pow)" % dt)
    if out is not None:
        assert out._dtype == dt
        flush_lazy_readers(out)
    else:
        out = HipTensor.empty(shape, dtype=dt)
    scalar = next((x for x in (a, b) if x is not None and not isinstance(x, HipTensor)), 0)
    ptrs = []
    for t in (a, b):
        ptrs += [t.ptr, i64(_bstrides(t, shape))] if isinstance(t, HipTensor) else [_NULL, _NULL]
    _l.check(_l.lib().lg_ew_typed(op, code, len(shape), i64(shape), out.ptr, i64(out._strides), *ptrs,
                                  float(scalar), int(scalar) if dt != _F64 else 0))
    return out


def _cast(t, dtype):
    """t.astype(dtype) as a new dense tensor (numpy's conversion rules: float -> int truncates toward zero)"""
    dtype = np.dtype(dtype)
    if dtype not in _CASTABLE or t._dtype not in _CASTABLE:
        raise TypeError("HipTensor.astype: %s -> %s (int16, int32, int64, float32, float64)" % (t._dtype, dtype))
    out = HipTensor.empty(t._shape, dtype=dtype, requires_grad=False)
    _l.check(_l.lib().lg_cast(_CASTABLE[t._dtype], _CASTABLE[dtype], len(t._shape), i64(t._shape), out.ptr, i64(out._strides),
                              t.ptr, i64(t._strides)))
    return out


def _unary(op, t):
    if t._dtype != _F32:
        return _ew_typed(op, t._dtype, t._shape, t, None)
    return _ew(op, t._shape, [t])


def _binary(op, a, b, out=None):
    """a (op) b with numpy broadcasting; either operand (not both) may be a python/numpy scalar"""
    dt = _operand_dtype(a, b)
    if dt != _F32 and dt is not None:
        scalar = b if _is_scalar(b) else (a if _is_scalar(a) else None)
        if dt != _F64 and scalar is not None and not isinstance(scalar, (int, np.integer)):
            # an integer tensor meets a float scalar: numpy computes in float64 (a python float is "weak" only among floats)
            if out is not None:
                raise TypeError("in-place operation of an %s tensor with a float scalar: the result is float64 and does not fit" % dt)
            a = _cast(a, _F64) if isinstance(a, HipTensor) else a
            b = _cast(b, _F64) if isinstance(b, HipTensor) else b
            dt = _F64
        tensors = [t for t in (a, b) if isinstance(t, HipTensor)]
        shape = tensors[0]._shape if len(tensors) == 1 or tensors[0]._shape == tensors[1]._shape else _broadcast_shapes(*(t._shape for t in tensors))
        return _ew_typed(op, dt, shape, a, b, out=out)
    if _is_scalar(b):
        return _ew(op, a._shape, [a, None], scalar=b, out=out)
    if _is_scalar(a):
        return _ew(op, b._shape, [None, b], scalar=a, out=out)
    shape = a._shape if a._shape == b._shape else _broadcast_shapes(a._shape, b._shape)
    return _ew(op, shape, [a, b], out=out)


def _rows(t, cols):
    """`t.reshape(-1, cols)` for the internals of an op (forward bodies and backward passes run with the tape switched off): a dense
    tensor becomes a view without a trip through the op dispatch - 6 us of host time per call, three calls per eager MLP step -
    anything else goes the general way"""
    if len(t._shape) == 2 and t._shape[1] == cols:
        return t
    if t._data is not None and t.is_contiguous() and cols > 0:
        return HipTensor(t._data, (t.numel() // cols, cols), None, t._offset, t._dtype, requires_grad=t._requires_grad)
    return t.reshape(-1, cols)


def _alias(t):
    """new tensor object on the same storage (what in-place ops return, cf. cpu/ops.py:120-146)"""
    return HipTensor(t.data, t._shape, t._strides, t._offset, t._dtype, requires_grad=t.requires_grad)


def _saved_output(y):
    """what a node keeps of its OWN output for the backward (exp, tanh, softmax, pow, max: the gradient is a function of y): a
    second tensor object on the same storage.  Saving y itself would close a reference cycle y -> ctx -> saved y, and the whole
    tape behind it would live until python's cycle collector happens to run - measured on the eager tiny-BERT training loop: the
    pool's reserved memory grew from 6.8 to 10.3 GB with 1 080 extra hipMalloc calls in 3 000 steps (tools/bert_train_soak.py)."""
    return HipTensor(y.data, y._shape, y._strides, y._offset, y._dtype, requires_grad=False)


""" Transformations """


@HipTensor.register_op()
@HipTensor.register_op("T")
class transpose(Function):
    """ stride permutation, no data movement (opencl/ops.py:9-27; numpy semantics cpu/ops.py:25-36) """
    def forward(ctx, a, *axes):
        if len(axes) == 0:
            axes = tuple(reversed(range(len(a._shape))))
        assert len(axes) == len(a._shape)
        axes = tuple(ax % len(axes) for ax in axes)
        ctx.save_for_backward(axes)
        return HipTensor(a.data, tuple(a._shape[i] for i in axes), tuple(a._strides[i] for i in axes), a._offset, a._dtype)

    def backward(ctx, out_grad):
        axes, = ctx.get_saved_tensors()
        inverse = [0] * len(axes)
        for i, j in enumerate(axes):
            inverse[j] = i
        return out_grad.transpose(*inverse)


@HipTensor.register_op()
class reshape(Function):
    """ view when the tensor is dense, gathered copy otherwise (opencl/ops.py:29-36, cpu/ops.py:38-47) """
    def forward(ctx, a, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        ctx.save_for_backward(a._shape)
        n = a.numel()
        if -1 in shape:
            known = 1
            for s in shape:
                if s != -1:
                    known *= s
            shape = tuple((n // known if known else 0) if s == -1 else s for s in shape)
        m = 1
        for s in shape:
            m *= s
        if m != n:
            raise ValueError("cannot reshape tensor of size %d into shape %s" % (n, shape))
        src = a.contiguous()
        out = HipTensor(src.data, shape, None, src._offset, a._dtype)
        if src is a and a.ctx is None and a.requires_grad:
            # a dense reshape of a LEAF (the `x.reshape(-1, 784)` of an MLP's input): gradients for the view may be added
            # straight into the leaf's gradient buffer, see HipTensor._grad_accumulator
            out._view_of_leaf = a
        return out

    def backward(ctx, out_grad):
        shape, = ctx.get_saved_tensors()
        return out_grad.reshape(*shape)


_tape_reshape = HipTensor.reshape


def _reshape_or_self(self, *shape):
    """`t.reshape(...)`.  Peephole: a dense tensor asked for the shape it already has (the `x.reshape(-1, 784)` in front of an MLP
    that is fed (batch, 784) rows) is returned as it is - no tape node, no dispatch (6 us of host time in the eager step, twice:
    the node also has a backward).  The reference returns a second tensor on the same array (cpu/ops.py:38-47); every use of it -
    values, in-place writes, gradients - is a use of the tensor itself."""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        shape = tuple(shape[0])
    if len(shape) == len(self._shape) and self._data is not None and self._dense is not False:
        same = True
        for want, have in zip(shape, self._shape):
            if want != have and want != -1:
                same = False
                break
        if same and shape.count(-1) <= 1 and self.is_contiguous():
            return self
    return _tape_reshape(self, *shape)


_reshape_or_self.__name__ = "reshape"
HipTensor.reshape = _reshape_or_self


""" Basic math """


@HipTensor.register_op()
class neg(Function):
    """ cpu/ops.py:52-58 """
    def forward(ctx, a):
        return _unary(_l.EW_NEG, a)

    def backward(ctx, out_grad):
        return _unary(_l.EW_NEG, out_grad)


@HipTensor.register_op()
class add(Function):
    """ cpu/ops.py:60-66 """
    def forward(ctx, a, b):
        return _binary(_l.EW_ADD, a, b)

    def backward(ctx, out_grad):
        return out_grad, out_grad


@HipTensor.register_op(overwrite=True)
class sub(Function):
    """ cpu/ops.py:68-74 """
    def forward(ctx, a, b):
        return _binary(_l.EW_SUB, a, b)

    def backward(ctx, out_grad):
        return out_grad, _unary(_l.EW_NEG, out_grad)


@HipTensor.register_op()
class mul(Function):
    """ cpu/ops.py:76-84; tensor*tensor backward is one two-output kernel (opencl/ops.py:78-83) """
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _binary(_l.EW_MUL, a, b)

    def backward(ctx, out_grad):
        a, b = ctx.get_saved_tensors()
        if _is_scalar(b):
            return _binary(_l.EW_MUL, out_grad, b)
        if _is_scalar(a):
            return None, _binary(_l.EW_MUL, out_grad, a)
        if out_grad._dtype != _F32:              # float64 tapes: the two products as two kernels (g * b, a * g: cpu/ops.py:83-84)
            return _binary(_l.EW_MUL, out_grad, b), _binary(_l.EW_MUL, a, out_grad)
        _require_f32(a, b, out_grad)
        shape = _broadcast_shapes(a._shape, b._shape, out_grad._shape)
        return _ew(_l.EW_MUL_BWD, shape, [a, b, out_grad], n_out=2)


@HipTensor.register_op(overwrite=True)
class div(Function):
    """ cpu/ops.py:86-94 """
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _binary(_l.EW_DIV, a, b)

    def backward(ctx, out_grad):
        a, b = ctx.get_saved_tensors()
        if _is_scalar(b):
            return _binary(_l.EW_DIV, out_grad, b)
        if _is_scalar(a):
            # -a / b**2 * g, evaluated left to right like the numpy expression
            t = _binary(_l.EW_DIV, -float(a), _binary(_l.EW_MUL, b, b))
            return None, _binary(_l.EW_MUL, t, out_grad)
        if out_grad._dtype != _F32:              # float64: g / b and -a / b**2 * g, left to right like cpu/ops.py:93-94
            return (_binary(_l.EW_DIV, out_grad, b),
                    _binary(_l.EW_MUL, _binary(_l.EW_DIV, _unary(_l.EW_NEG, a), _binary(_l.EW_MUL, b, b)), out_grad))
        _require_f32(a, b, out_grad)
        shape = _broadcast_shapes(a._shape, b._shape, out_grad._shape)
        return _ew(_l.EW_DIV_BWD, shape, [a, b, out_grad], n_out=2)


def _pow_scalar(a, e):
    """a ** e for a scalar exponent, with numpy's scalar fast paths (square, sqrt, reciprocal, identity)"""
    e = float(e)
    if e == 2.0:
        return _binary(_l.EW_MUL, a, a)
    if e == 1.0:
        return _unary(_l.EW_COPY, a)
    if e == 0.5:
        return _unary(_l.EW_SQRT, a)
    if e == -1.0:
        return _binary(_l.EW_DIV, 1.0, a)
    return _binary(_l.EW_POW, a, e)


@HipTensor.register_op()
class pow(Function):
    """ cpu/ops.py:96-105 """
    def forward(ctx, a, b):
        y = _pow_scalar(a, b) if _is_scalar(b) else _binary(_l.EW_POW, a, b)
        ctx.save_for_backward(a, b, _saved_output(y))
        return y

    def backward(ctx, out_grad):
        a, b, y = ctx.get_saved_tensors()
        if _is_scalar(b):
            # b * a**(b-1) * g
            return _binary(_l.EW_MUL, _binary(_l.EW_MUL, _pow_scalar(a, float(b) - 1.0), b), out_grad)
        if _is_scalar(a):
            # g * y * log(a)
            return None, _binary(_l.EW_MUL, _binary(_l.EW_MUL, out_grad, y), float(np.log(np.float32(a))))
        _require_f32(a, b, out_grad)
        shape = _broadcast_shapes(a._shape, b._shape, out_grad._shape)
        return _ew(_l.EW_POW_BWD, shape, [a, b, out_grad, y], n_out=2)


""" Matrix product """


class _Mat(object):
    """2-D (optionally batched) operand as the GEMM sees it: pointer + which index is contiguous + ld"""
    __slots__ = ("t", "rows", "cols", "colmajor", "ld")

    def __init__(self, t, rows, cols, sr, sc):
        if cols == 1 or sc == 1:
            self.colmajor, self.ld, self.t = False, _py.max(sr, cols) if rows > 1 else cols, t
        elif rows == 1 or sr == 1:
            self.colmajor, self.ld, self.t = True, _py.max(sc, rows) if cols > 1 else rows, t
        else:
            raise ValueError("not a GEMM operand layout")
        self.rows, self.cols = rows, cols


def _as_mat(t):
    """(tensor to keep alive, _Mat) for the last two dims of t; copies only if neither dim is unit-stride"""
    rows, cols = t._shape[-2], t._shape[-1]
    sr, sc = t._strides[-2], t._strides[-1]
    ok = (cols == 1 or sc == 1) and (rows == 1 or sr >= cols) or (rows == 1 or sr == 1) and (cols == 1 or sc >= rows)
    if not ok:
        t = t.contiguous()
        sr, sc = t._strides[-2], t._strides[-1]
    return _Mat(t, rows, cols, sr, sc)


def _collapse_batch(shape, strides):
    """(count, stride) if the batch dims walk as one strided run, else None"""
    dims = [(s, st) for s, st in zip(shape, strides) if s != 1]
    if not dims:
        return 1, 0
    n, st = dims[-1]
    for s, sst in reversed(dims[:-1]):
        if sst != st * n:
            return None
        n *= s
    return n, st


def _dense_permutation(shape, strides) -> bool:
    """do `strides` address a dense block of prod(shape) elements, each exactly once (a stride-permuted dense tensor)?"""
    expect = 1
    for extent, stride in sorted(((e, st) for e, st in zip(shape, strides) if e != 1), key=lambda d: d[1]):
        if stride != expect:
            return False
        expect *= extent
    return True


def _layout_like(t):
    """strides for a fresh tensor of t's shape laid out in memory like t - when t is a stride-permuted dense tensor (the head
    split `(b, s, h*d) -> (b, h, s, d)` of attention) that a GEMM can store into (one of the last two dims has stride 1);
    None for dense t (nothing to imitate) and for anything else"""
    if len(t._shape) < 2 or t.is_contiguous() or not _dense_permutation(t._shape, t._strides):
        return None
    if not (t._strides[-1] == 1 or t._shape[-1] == 1 or t._strides[-2] == 1 or t._shape[-2] == 1):
        return None
    return t._strides


def _product_layout(a, b, out_shape):
    """Memory layout for a @ b when the right operand is a head-split view with rows of N contiguous values - `probs @ v`
    with v = (b, s, h*d) seen as (b, h, s, d): the product takes v's nesting of batch and row dims, (b, s_q, h, d) in memory,
    so that the `transpose(0, 2, 1, 3).reshape(b, s, h*d)` that follows (bert.py:87) is a view instead of a copy.  None: dense."""
    if len(b._shape) < 3 or a._shape[:-2] != b._shape[:-2] or b._strides[-1] != 1 or b.is_contiguous() \
            or not _dense_permutation(b._shape, b._strides):
        return None
    # dims of b ordered from the innermost in memory; the product's dim -2 (M) takes the place of b's dim -2 (K)
    order = sorted(range(len(b._shape)), key=lambda i: (b._strides[i], -i))
    strides, run = [0] * len(out_shape), 1
    for i in order:
        strides[i] = run
        run *= out_shape[i]
    return tuple(strides)


def _gemm(a, b, out_colmajor=False, bias=None, accumulate_into=None, overwrite=False, out_strides=None, addend=None):
    """a (..., M, K) @ b (..., K, N) [+ bias (N,)] -> (..., M, N) on the MFMA SGEMM kernel.

    Operands are consumed in place whenever one of their last two dims has stride 1
    (row-major or stride-permuted views alike).  `out_colmajor` stores the result
    transposed in memory and returns the matching view, `out_strides` (a dense permutation
    with a unit stride among the last two dims, see _layout_like) stores it in any such
    layout - so that a gradient can be produced directly in the layout of the tensor it
    belongs to and an attention product in the layout its consumer reshapes for free.
    `addend` (dense, the shape of the result; one matrix product only) is added last:
    (a @ b + bias) + addend from the GEMM's epilogue.
    """
    _require_f32(a, b)
    squeeze_a = squeeze_b = False
    if len(a._shape) == 1:
        a, squeeze_a = a.reshape(1, a._shape[0]), True
    if len(b._shape) == 1:
        b, squeeze_b = b.reshape(b._shape[0], 1), True
    M, K = a._shape[-2], a._shape[-1]
    K2, N = b._shape[-2], b._shape[-1]
    if K != K2:
        raise ValueError("matmul: shapes %s and %s do not align" % (a._shape, b._shape))
    batch_shape = _broadcast_shapes(a._shape[:-2], b._shape[:-2]) if (len(a._shape) > 2 or len(b._shape) > 2) else ()

    # (B..., M, K) @ (K, N) with dense leading dims is ONE tall GEMM
    if len(b._shape) == 2 and len(a._shape) > 2:
        assert out_strides is None
        lead = _collapse_batch(a._shape[:-1], a._strides[:-1])
        if lead is None or not (a._strides[-1] == 1 or K == 1):
            a = a.contiguous()
            lead = _collapse_batch(a._shape[:-1], a._strides[:-1])
        rows, rstride = lead
        flat = HipTensor(a.data, (rows, K), (rstride if rows > 1 else K, a._strides[-1]), a._offset, a._dtype)
        if addend is not None:
            assert addend._shape == batch_shape + (M, N) and addend.is_contiguous()
            addend = HipTensor(addend.data, (rows, N), None, addend._offset, addend._dtype)
        out = _gemm(flat, b, out_colmajor=False, bias=bias, addend=addend)
        return out.reshape(*batch_shape, M, N)

    ma, mb = _as_mat(a), _as_mat(b)
    a, b = ma.t, mb.t
    nb = 1
    for s in batch_shape:
        nb *= s
    out_shape = batch_shape + (M, N)
    dense = contiguous_strides(out_shape)
    if out_colmajor:
        assert out_strides is None
        out_strides = dense[:-2] + (1, M)
    if out_strides is not None and tuple(out_strides) == dense:
        out_strides = None
    if accumulate_into is not None:
        # C += A @ B straight into an existing dense buffer (a parameter's gradient): no temporary, no add pass
        assert out_strides is None and bias is None and not batch_shape
        assert accumulate_into._shape == out_shape and accumulate_into.is_contiguous() and accumulate_into._dtype == _F32
        flush_lazy_readers(accumulate_into)
        out = accumulate_into
    elif out_strides is not None:
        assert bias is None and not (squeeze_a or squeeze_b) and len(out_strides) == len(out_shape)
        assert _dense_permutation(out_shape, out_strides), "matmul: output layout %s is not a dense permutation of %s" % (out_strides, out_shape)
        out = HipTensor(HipBuffer(nb * M * N * 4), out_shape, tuple(out_strides), 0, _F32)
    else:
        out = HipTensor.empty(out_shape)
    # how the kernel stores one (M, N) result: rows of N with ldc between them, or - for a layout whose M index is the
    # contiguous one - the transposed product C^T = B^T @ A^T as rows of M
    s_m, s_n = out._strides[-2], out._strides[-1]
    if N == 1 or s_n == 1:
        t_store, ldc = False, (s_m if M > 1 else N)
    else:
        assert M == 1 or s_m == 1, "matmul: output layout needs a unit stride among its last two dims"
        t_store, ldc = True, (s_n if N > 1 else M)
    assert ldc >= (M if t_store else N)

    def batch_layout(t):
        bs = _bstrides(HipTensor(t.data, t._shape[:-2], t._strides[:-2], t._offset, t._dtype), batch_shape) if batch_shape else ()
        return bs

    sa, sb = batch_layout(a), batch_layout(b)
    so = tuple(out._strides[:-2])

    L = _l.lib()

    if bias is not None:
        assert not t_store and bias._shape == (N,) and bias.is_contiguous() and bias._dtype == _F32

    if addend is not None:
        assert not batch_shape and not t_store and accumulate_into is None and K > 0
        assert addend._shape == (M, N) and addend.is_contiguous() and addend._dtype == _F32

    def launch(pa, pb, po, count, stra, strb, stro):
        if addend is not None:
            _l.check(L.lg_gemm_addend_f32(1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K, pa, ma.ld, pb, mb.ld, po, ldc,
                                          bias.ptr if bias is not None else None, addend.ptr, N))
        elif bias is not None:
            _l.check(L.lg_gemm_bias_f32(1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K,
                                        pa, ma.ld, stra, pb, mb.ld, strb, po, ldc, stro, count, bias.ptr))
        elif t_store:
            # C^T (N x M, row-major) = B^T @ A^T : swap the operands and flip their layouts
            _l.check(L.lg_gemm_f32(0 if mb.colmajor else 1, 0 if ma.colmajor else 1, N, M, K,
                                   pb, mb.ld, strb, pa, ma.ld, stra, po, ldc, stro, count, 0))
        else:
            _l.check(L.lg_gemm_f32(1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K,
                                   pa, ma.ld, stra, pb, mb.ld, strb, po, ldc, stro, count,
                                   1 if (accumulate_into is not None and not overwrite) else 0))

    if nb > 0 and M > 0 and N > 0:
        ca, cb, co = _collapse_batch(batch_shape, sa), _collapse_batch(batch_shape, sb), _collapse_batch(batch_shape, so)
        outer = None
        if bias is None and len(batch_shape) >= 2:
            outer = [_collapse_batch(batch_shape[:-1], st[:-1]) for st in (sa, sb, so)]
        if ca is not None and cb is not None and co is not None:
            launch(a.ptr, b.ptr, out.ptr, nb, ca[1], cb[1], co[1])
        elif outer is not None and None not in outer:
            # attention after the head split: (batch, head) do not walk as ONE stride, but as two - still one launch
            (n_outer, a_outer), (_, b_outer), (_, o_outer) = outer
            inner = batch_shape[-1]
            if t_store:      # C^T = B^T @ A^T, as in launch()
                _l.check(L.lg_gemm_batched2_f32(0 if mb.colmajor else 1, 0 if ma.colmajor else 1, N, M, K,
                                                b.ptr, mb.ld, b_outer, sb[-1], a.ptr, ma.ld, a_outer, sa[-1],
                                                out.ptr, ldc, o_outer, so[-1], n_outer, inner, 0))
            else:
                _l.check(L.lg_gemm_batched2_f32(1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K,
                                                a.ptr, ma.ld, a_outer, sa[-1], b.ptr, mb.ld, b_outer, sb[-1],
                                                out.ptr, ldc, o_outer, so[-1], n_outer, inner, 0))
        else:
            # batch dims that do not collapse at all: loop over the leading dims, batch the innermost one
            inner = batch_shape[-1]
            for idx in np.ndindex(*batch_shape[:-1]):
                oa = _py.sum(i * s for i, s in zip(idx, sa[:-1])) * 4
                ob = _py.sum(i * s for i, s in zip(idx, sb[:-1])) * 4
                oo = _py.sum(i * s for i, s in zip(idx, so[:-1])) * 4
                launch(a.ptr + oa, b.ptr + ob, out.ptr + oo, inner, sa[-1], sb[-1], so[-1])
    if squeeze_a:
        out = out.reshape(*out._shape[:-2], out._shape[-1])
    if squeeze_b:
        out = out.reshape(*out._shape[:-1])
    return out


def _gemm_rowsum(a, b, accumulate_into=None, overwrite=False, rowsum_into=None, rowsum_overwrite=False):
    """(a @ b, a.sum(axis=1)) for 2-D operands from ONE launch (lg_gemm_rowsum_f32): the row sums are the product with a
    virtual column of ones.  With a = g^T, b = x this is nn.Linear's (dW, db).  Either result may be added into / written
    over an existing dense buffer (a parameter's gradient)."""
    _require_f32(a, b)
    (M, K), (K2, N) = a._shape, b._shape
    assert K == K2 and M > 0 and N > 0
    ma, mb = _as_mat(a), _as_mat(b)
    for t, shape in ((accumulate_into, (M, N)), (rowsum_into, (M,))):
        assert t is None or (t._shape == shape and t.is_contiguous() and t._dtype == _F32)
    for t in (accumulate_into, rowsum_into):
        if t is not None:
            flush_lazy_readers(t)
    out = accumulate_into if accumulate_into is not None else HipTensor.empty((M, N))
    rowsum = rowsum_into if rowsum_into is not None else HipTensor.empty((M,))
    _l.check(_l.lib().lg_gemm_rowsum_f32(1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K, ma.t.ptr, ma.ld, mb.t.ptr, mb.ld,
                                         out.ptr, N, 1 if (accumulate_into is not None and not overwrite) else 0,
                                         rowsum.ptr, 1 if (rowsum_into is not None and not rowsum_overwrite) else 0))
    return out, rowsum


def _gemm_fused(a, b, bias=None, accumulate_into=None, overwrite=False, rowsum_into=None, rowsum_overwrite=False,
                want_rowsum=False, relu_a=False, relu_b=False):
    """2-D a @ b through lg_gemm_fused_f32: optional bias, accumulation into an existing buffer, row sums of a, relu on
    either operand on the fly.  Returns (out, rowsum|None)."""
    _require_f32(a, b)
    (M, K), (K2, N) = a._shape, b._shape
    assert K == K2 and M > 0 and N > 0
    ma, mb = _as_mat(a), _as_mat(b)
    for t in (accumulate_into, rowsum_into):
        if t is not None:
            flush_lazy_readers(t)
    out = accumulate_into if accumulate_into is not None else HipTensor.empty((M, N))
    assert out._shape == (M, N) and out.is_contiguous()
    rowsum = None
    if want_rowsum or rowsum_into is not None:
        rowsum = rowsum_into if rowsum_into is not None else HipTensor.empty((M,))
        assert rowsum._shape == (M,) and rowsum.is_contiguous()
    if bias is not None:
        assert bias._shape == (N,) and bias.is_contiguous() and bias._dtype == _F32 and accumulate_into is None and rowsum is None
    _l.check(_l.lib().lg_gemm_fused_f32(
        1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K, ma.t.ptr, ma.ld, mb.t.ptr, mb.ld, out.ptr, N,
        1 if (accumulate_into is not None and not overwrite) else 0, bias.ptr if bias is not None else None,
        rowsum.ptr if rowsum is not None else None, 1 if (rowsum_into is not None and not rowsum_overwrite) else 0,
        1 if relu_a else 0, 1 if relu_b else 0))
    return out, rowsum


def _gemm_act(a, b, act, aux=None, bias=None):
    """2-D a @ b through lg_gemm_act_f32.  act = ACT_GELU: returns (pre, gelu(pre)) with pre = a @ b + bias.
    act = ACT_GELU_BWD: returns ((a @ b) * gelu'(aux), None), aux the pre-activation the forward kept."""
    _require_f32(a, b)
    (M, K), (K2, N) = a._shape, b._shape
    assert K == K2 and M > 0 and N > 0 and K > 0
    ma, mb = _as_mat(a), _as_mat(b)
    out = HipTensor.empty((M, N))
    if act == _l.ACT_GELU:
        aux = HipTensor.empty((M, N))
        assert bias is None or (bias._shape == (N,) and bias.is_contiguous() and bias._dtype == _F32)
    else:
        assert bias is None and aux._shape == (M, N) and aux.is_contiguous() and aux._dtype == _F32
    _l.check(_l.lib().lg_gemm_act_f32(1 if ma.colmajor else 0, 1 if mb.colmajor else 0, M, N, K, ma.t.ptr, ma.ld, mb.t.ptr, mb.ld,
                                      out.ptr, N, bias.ptr if bias is not None else None, act, aux.ptr, N))
    return out, (aux if act == _l.ACT_GELU else None)


def _lazy_relu_input(x):
    """the pre-activation t if x is a still-lazy relu(t) that a 2-D GEMM can read in its place, else None"""
    if x._data is None and x._lazy_source is not None and x._lazy_source[0] == "relu" and len(x._shape) == 2:
        return x._lazy_source[1]
    return None


def _is_colmajor(t):
    return len(t._shape) >= 2 and t._strides[-2] == 1 and t._shape[-1] > 1 and t._strides[-1] != 1


def _swap_last(t):
    return HipTensor(t.data, t._shape[:-2] + (t._shape[-1], t._shape[-2]), t._strides[:-2] + (t._strides[-1], t._strides[-2]),
                     t._offset, t._dtype)


@HipTensor.register_op()
@HipTensor.register_op("__matmul__")
class dot(Function):
    """ a @ b (cpu/ops.py:107-116); backward g @ b^T, a^T @ g with the last two axes swapped so it is
    also right for batched operands (opencl/ops.py:127-132).  Transposed operands are views; each
    gradient is produced in the memory layout of the operand it belongs to, so `W.T(1,0)` in
    nn.Linear gets a gradient that transposes back to a dense dW without a strided accumulate. """
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        if len(a._shape) > 2 and len(b._shape) > 2:
            return _gemm(a, b, out_strides=_product_layout(a, b, _broadcast_shapes(a._shape[:-2], b._shape[:-2]) + (a._shape[-2], b._shape[-1])))
        return _gemm(a, b)

    def backward(ctx, out_grad):
        a, b = ctx.get_saved_tensors()
        if len(a._shape) < 2 or len(b._shape) < 2:
            # vector operands: fall back to explicit 2-D views
            a2 = a.reshape(1, -1) if len(a._shape) == 1 else a
            b2 = b.reshape(-1, 1) if len(b._shape) == 1 else b
            g2 = out_grad.reshape(*a2._shape[:-1], b2._shape[-1])
            ga = _gemm(g2, _swap_last(b2)).reshape(*a._shape)
            gb = _gemm(_swap_last(a2), g2).reshape(*b._shape)
            return ga, gb
        if len(b._shape) == 2 and len(a._shape) > 2:
            # (B..., M, K) @ (K, N): dB = a_flat^T @ g_flat in one GEMM (no per-batch products + reduction)
            K, N = b._shape
            a_flat, g_flat = a.reshape(-1, K), out_grad.reshape(-1, N)
            ga = _gemm(g_flat, _swap_last(b), out_colmajor=False).reshape(*a._shape)
            gb = _gemm(_swap_last(a_flat), g_flat, out_colmajor=_is_colmajor(b))
            return ga, gb
        ga = gb = None
        if len(a._shape) == 2 and len(b._shape) == 2:
            # dense 2-D leaves that already own a gradient buffer: accumulate in the GEMM epilogue (see linear.backward)
            acc_a, acc_b = a._grad_accumulator(), b._grad_accumulator()
            if a.requires_grad and acc_a is not None and acc_a.is_contiguous() and a is not b:
                _gemm(out_grad, _swap_last(b), accumulate_into=acc_a, overwrite=a._consume_zero_pending())
                a._notify_grad_written()
                ga = False
            if b.requires_grad and acc_b is not None and acc_b.is_contiguous() and a is not b:
                _gemm(_swap_last(a), out_grad, accumulate_into=acc_b, overwrite=b._consume_zero_pending())
                b._notify_grad_written()
                gb = False
        # each gradient in the memory layout of the operand it belongs to (a head-split view, a transposed matrix): the
        # transpose / reshape backward that follow are then views, not gathered copies
        same_batch = a._shape[:-2] == b._shape[:-2] == out_grad._shape[:-2]

        def grad_a():
            like = _layout_like(a) if same_batch else None
            return _gemm(out_grad, _swap_last(b), out_strides=like) if like is not None else _gemm(out_grad, _swap_last(b), out_colmajor=_is_colmajor(a))

        def grad_b():
            like = _layout_like(b) if same_batch else None
            return _gemm(_swap_last(a), out_grad, out_strides=like) if like is not None else _gemm(_swap_last(a), out_grad, out_colmajor=_is_colmajor(b))

        if ga is None and gb is None and len(a._shape) > 2 and out_grad.numel() > 0:
            # attention's two gradients per product (dS^T @ q with dS @ k;  probs^T @ dO with dO @ v^T): independent, small, and in
            # the layouts the two-product launch takes - one launch instead of two (lg_gemm_pair_*; the M-contiguous-A product
            # has to be offered first).  Whatever does not fit the pair is launched on its own, as without the bracket.
            _l.check(_l.lib().lg_gemm_pair_begin())
            try:
                gb = grad_b()
                ga = grad_a()
            finally:
                _l.check(_l.lib().lg_gemm_pair_end())
        if ga is None:
            ga = grad_a()
        if gb is None:
            gb = grad_b()
        return (None if ga is False else ga), (None if gb is False else gb)


""" In-place operators: no backward; the result aliases the input storage (cpu/ops.py:120-153) """


def _inplace(op, t, other):
    if isinstance(other, HipTensor):
        assert _broadcast_shapes(t._shape, other._shape) == t._shape, \
            "in-place operand of shape %s does not broadcast to %s" % (other._shape, t._shape)
    _binary(op, t, other, out=t)
    return _alias(t)


@HipTensor.register_op()
class astype(Function):
    """ t.astype(dtype): numpy's conversion (the reference has no cast op: its tensors change dtype through numpy on the host,
    cpu/tensor.py:12).  Differentiable between float dtypes: the gradient is cast back. """
    def forward(ctx, t, dtype=np.float32):
        ctx.save_for_backward(t._dtype, np.dtype(dtype))
        out = _cast(t, dtype) if np.dtype(dtype) != t._dtype else _unary(_l.EW_COPY, t)
        out._requires_grad = True             # (whether it joins the tape is the Function machinery's call, from the input)
        return out

    def backward(ctx, out_grad):
        src, dst = ctx.get_saved_tensors()
        if src.kind != "f" or dst.kind != "f":
            raise RuntimeError("Cannot Backward through astype(%s -> %s)!" % (src, dst))
        return _cast(out_grad, src) if out_grad._dtype != src else out_grad


@HipTensor.register_op("__iadd__", overwrite=True)
class iadd(Function):
    def forward(ctx, t, other):
        return _inplace(_l.EW_ADD, t, other)


@HipTensor.register_op("__isub__", overwrite=True)
class isub(Function):
    def forward(ctx, t, other):
        return _inplace(_l.EW_SUB, t, other)


@HipTensor.register_op("__imul__", overwrite=True)
class imul(Function):
    def forward(ctx, t, other):
        return _inplace(_l.EW_MUL, t, other)


@HipTensor.register_op("__itruediv__", overwrite=True)
class itruediv(Function):
    def forward(ctx, t, other):
        return _inplace(_l.EW_DIV, t, other)


def _value_bits(val, dtype):
    raw = np.asarray(val).astype(dtype).tobytes()
    return int.from_bytes(raw, "little")


@HipTensor.register_op()
class fill(Function):
    """ opencl/ops.py:172-177; works on strided views and any dtype """
    def forward(ctx, t, val):
        flush_lazy_readers(t)
        _l.check(_l.lib().lg_fill_strided(t._dtype.itemsize, len(t._shape), i64(t._shape), t.ptr, i64(t._strides),
                                          _value_bits(val, t._dtype)))
        return t


""" Non-linearities """


def _unary_op(name, fwd, bwd, save_output, cite):
    class Op(Function):
        def forward(ctx, t):
            y = _unary(fwd, t)
            ctx.save_for_backward(_saved_output(y) if save_output else t)
            return y

        def backward(ctx, out_grad):
            s, = ctx.get_saved_tensors()
            return _binary(bwd, s, out_grad)
    Op.__name__ = Op.__qualname__ = name
    Op.__doc__ = cite
    return Op


# exp saves its output (y * g), sigmoid/tanh too; sin/cos/log/relu save the input
exp = HipTensor.register_op("exp", _unary_op("exp", _l.EW_EXP, _l.EW_MUL, True, "cpu/ops.py:178-187"))
log = HipTensor.register_op("log", _unary_op("log", _l.EW_LOG, _l.EW_LOG_BWD, False, "cpu/ops.py:189-197"))
sin = HipTensor.register_op("sin", _unary_op("sin", _l.EW_SIN, _l.EW_SIN_BWD, False, "cpu/ops.py:158-166"))
cos = HipTensor.register_op("cos", _unary_op("cos", _l.EW_COS, _l.EW_COS_BWD, False, "cpu/ops.py:168-176"))
sigmoid = HipTensor.register_op("sigmoid", _unary_op("sigmoid", _l.EW_SIGMOID, _l.EW_SIGMOID_BWD, True, "cpu/ops.py:199-208"),
                                overwrite=True)
tanh = HipTensor.register_op("tanh", _unary_op("tanh", _l.EW_TANH, _l.EW_TANH_BWD, True, "cpu/ops.py:210-219"), overwrite=True)
@HipTensor.register_op()
class relu(Function):
    """ np.maximum(t, 0); gradient passes at exactly 0: g * (t >= 0) (cpu/ops.py:221-229).
    The result is LAZY for dense inputs (HipTensor._lazy_source): `linear` folds the relu into its GEMMs - as forward
    operand and as weight-gradient operand - so that in `Linear -> relu -> Linear` the relu kernel never runs and its
    output is never written; every other consumer makes the tensor real by asking for its data. """
    def forward(ctx, t):
        _require_f32(t)
        ctx.save_for_backward(t)
        if t.is_contiguous() and t.numel() > 0:
            out = HipTensor(None, t._shape, None, 0, t._dtype)
            out._lazy_source = ("relu", t)
            out._watch_sources(t)      # whoever writes into t's storage before `out` exists computes it first
            return out
        return _unary(_l.EW_RELU, t)

    def backward(ctx, out_grad):
        t, = ctx.get_saved_tensors()
        done = out_grad._relu_bwd_done
        if done is not None and done[0] is t:
            # the kernel that made this gradient (head_bwd) already wrote `out_grad * (t >= 0)` next to it
            return done[1]
        return _binary(_l.EW_RELU_BWD, t, out_grad)


""" Selectors """


_INT_INDEX_DTYPES = (np.dtype(np.int16), np.dtype(np.int32), np.dtype(np.int64))


def _is_advanced(i):
    return isinstance(i, (list, range, HipTensor)) or (isinstance(i, np.ndarray) and i.ndim > 0)


class _TakePlan(object):
    """integer-array indexing split into a basic view + ONE gather along an axis of that view (csrc/index.hip).  `merge` > 1:
    the gather runs along that many neighbouring axes of the (dense) view taken as one axis, with row-major flat indices"""
    __slots__ = ("basic", "axis", "index", "pair", "merge", "perm")


def _is_bool_index(i):
    return (isinstance(i, HipTensor) and i._dtype == np.dtype(np.bool_)) or (isinstance(i, np.ndarray) and i.dtype == np.bool_ and i.ndim > 0) \
        or (isinstance(i, list) and len(i) > 0 and isinstance(i[0], (bool, np.bool_)))


def _index_tensor(x, axis_len):
    """an index given as HipTensor / ndarray / list / range -> dense int HipTensor on the device.  Host-side indices are
    validated here (IndexError at once, like numpy); device-resident ones by the kernel (status flag, next sync)."""
    if isinstance(x, HipTensor):
        if x._dtype not in _INT_INDEX_DTYPES:
            raise IndexError("index tensors must be int16, int32 or int64 (got %s); boolean masks are not supported" % x._dtype)
        return x.contiguous()
    arr = np.asarray(x)
    if arr.dtype.kind not in "iu":
        raise IndexError("arrays used as indices must be of integer type (got %s)" % arr.dtype)
    arr = arr.astype(np.int64)
    if arr.size and (arr.min() < -axis_len or arr.max() >= axis_len):
        bad = arr[(arr < -axis_len) | (arr >= axis_len)].flat[0]
        raise IndexError("index %d is out of bounds for axis with size %d" % (bad, axis_len))
    return HipTensor.from_numpy(arr, requires_grad=False)


def _is_arange(x, n):
    if isinstance(x, range):
        return len(x) == n and (n == 0 or (x[0] == 0 and x.step == 1))
    if isinstance(x, HipTensor):
        return False
    arr = np.asarray(x)
    return arr.ndim == 1 and arr.shape[0] == n and arr.dtype.kind in "iu" and np.array_equal(arr, np.arange(n))


def _mask_positions(mask, dims):
    """flat row-major positions of the True elements of a boolean mask over axes of lengths `dims` (numpy: x[mask] ==
    x[mask.nonzero()]), as a dense int64 device tensor of shape (count,).  A host mask is searched on the host; a device mask on
    the device (lg_mask_nonzero) - only the COUNT comes back, because the result's shape depends on it."""
    shape = mask._shape if isinstance(mask, HipTensor) else np.asarray(mask).shape
    if tuple(shape) != tuple(dims):
        raise IndexError("boolean index did not match indexed array: mask of shape %s on dimensions of shape %s" % (tuple(shape), tuple(dims)))
    if not isinstance(mask, HipTensor):
        flat = np.flatnonzero(np.asarray(mask, dtype=np.bool_))
        return HipTensor.from_numpy(flat.astype(np.int64), requires_grad=False)
    m = mask.contiguous()
    n = m.numel()
    positions = HipTensor.empty((_py.max(n, 1),), dtype=np.int64, requires_grad=False)
    count = HipTensor.empty((1,), dtype=np.int64, requires_grad=False)
    _l.check(_l.lib().lg_mask_nonzero(m.ptr, n, positions.ptr, count.ptr))
    k = int(count.numpy()[0])                                   # the one read-back: the length of the result
    return HipTensor(positions.data, (k,), None, positions._offset, positions._dtype, requires_grad=False)


def _fold_indices(groups, bshape):
    """ONE flat row-major index over the axes the advanced indices of a subscript cover, on the device (lg_index_fold).
    groups: [(length the index runs over, dense device index tensor | python int)]; bshape: the arrays' broadcast shape"""
    k, nd = len(groups), len(bshape)
    out = HipTensor.empty(bshape, dtype=np.int64, requires_grad=False)
    lens = i64(tuple(int(g[0]) for g in groups))
    ptrs = (ctypes.c_void_p * k)(*[g[1].ptr if isinstance(g[1], HipTensor) else None for g in groups])
    sizes = (ctypes.c_int * k)(*[g[1]._dtype.itemsize if isinstance(g[1], HipTensor) else 8 for g in groups])
    consts = i64(tuple(0 if isinstance(g[1], HipTensor) else int(g[1]) for g in groups))
    strides = [i64(_bstrides(g[1], bshape)) if isinstance(g[1], HipTensor) else i64((0,) * _py.max(nd, 1)) for g in groups]
    stride_ptrs = (ctypes.POINTER(ctypes.c_int64) * k)(*[ctypes.cast(st, ctypes.POINTER(ctypes.c_int64)) for st in strides])
    _l.check(_l.lib().lg_index_fold(k, lens, ptrs, sizes, consts, nd, i64(tuple(bshape)), stride_ptrs, out.ptr))
    return out


def _take_plan(a, idx):
    """None for basic indexing, else the _TakePlan of `a[idx]`: what numpy does with a subscript that holds index arrays or
    boolean masks (cpu/ops.py:234-255 hands it to numpy).  ONE index array next to ints / slices and the pair `range(n), labels`
    (loss.py:19) gather straight along their axis; everything else - several arrays broadcast together, masks over any run of
    axes, plain integers among them, arrays on NON-neighbouring axes (numpy then moves the index dimensions to the front) - is
    folded on the device into one flat index over the covered axes, which are brought together by a transposed view first."""
    idx = idx if isinstance(idx, tuple) else (idx,)
    if not any(_is_advanced(i) for i in idx):
        return None
    n_real = 0
    for i in idx:
        if i is None or i is Ellipsis:
            continue
        n_real += len(i._shape if isinstance(i, HipTensor) else np.asarray(i).shape) if _is_advanced(i) and _is_bool_index(i) else 1
    if any(i is Ellipsis for i in idx):             # (`Ellipsis in idx` would compare the index arrays with ==)
        k = next(k for k, i in enumerate(idx) if i is Ellipsis)
        idx = idx[:k] + (slice(None),) * (len(a._shape) - n_real) + idx[k + 1:]
    # where every entry lands: axis of `a` it starts at, kind
    arrays = [i for i in idx if _is_advanced(i) and not _is_bool_index(i)]
    masks = [i for i in idx if _is_advanced(i) and _is_bool_index(i)]
    has_none = any(i is None for i in idx)
    ints_count = [i for i in idx if not (i is None or isinstance(i, slice) or _is_advanced(i))]
    # positions (among the non-None entries) of everything numpy treats as an advanced index here: arrays, masks and plain ints
    order = [i for i in idx if i is not None]
    adv_pos = [k for k, i in enumerate(order) if _is_advanced(i) or not isinstance(i, slice)]
    adjacent = adv_pos[-1] - adv_pos[0] == len(adv_pos) - 1

    plan = _TakePlan()
    plan.merge, plan.pair, plan.perm = 1, False, None
    if len(arrays) == 1 and not masks and (adjacent or not ints_count):
        # one index array; ints next to it just drop their axes first (the result is the same as numpy's in-place rule)
        plan.basic = tuple(slice(None) if _is_advanced(i) else i for i in idx)
        axis = 0
        for i in idx[:next(k for k, i in enumerate(idx) if _is_advanced(i))]:
            if i is None or isinstance(i, slice):
                axis += 1
        plan.axis = axis
        plan.index = _index_tensor(arrays[0], _idx_view(a, plan.basic)._shape[axis])
        return plan
    if has_none:
        raise IndexError("HipTensor indexing: None (newaxis) together with several index arrays / masks is not supported - index first, reshape then")
    first = next(k for k, i in enumerate(idx) if _is_advanced(i))
    if (len(arrays) == 2 and not masks and not ints_count and adjacent):
        view_shape = _idx_view(a, tuple(slice(None) if _is_advanced(i) else i for i in idx))._shape
        axis = _py.sum(1 for i in idx[:first] if isinstance(i, slice))
        if _is_arange(arrays[0], view_shape[axis]):
            n = view_shape[axis]
            plan.basic = tuple(slice(None) if _is_advanced(i) else i for i in idx)
            plan.axis, plan.pair = axis, True
            plan.index = _index_tensor(arrays[1], view_shape[axis + 1])
            if plan.index._shape != (n,):
                raise IndexError("shape mismatch: indexing arrays could not be broadcast together with shapes (%d,) %s" % (n, plan.index._shape))
            return plan
    # ---- the general form.  Every advanced entry (array, mask, plain int) keeps its axes in the basic view; `groups` lists them
    basic, groups, adv_axes, axis = [], [], [], 0
    for i in idx:
        if isinstance(i, slice):
            basic.append(i)
            axis += 1
        elif _is_advanced(i) and _is_bool_index(i):
            m = len(i._shape if isinstance(i, HipTensor) else np.asarray(i).shape)
            dims = a._shape[axis:axis + m]
            length = 1
            for dlen in dims:
                length *= dlen
            groups.append((length, _mask_positions(i, dims)))
            adv_axes += list(range(axis, axis + m))
            basic += [slice(None)] * m
            axis += m
        elif _is_advanced(i):
            groups.append((a._shape[axis], _index_tensor(i, a._shape[axis])))
            adv_axes.append(axis)
            basic.append(slice(None))
            axis += 1
        else:
            size, v = a._shape[axis], int(i)
            if v < -size or v >= size:
                raise IndexError("index %d is out of bounds for axis %d with size %d" % (v, axis, size))
            groups.append((size, v))
            adv_axes.append(axis)
            basic.append(slice(None))
            axis += 1
    if axis > len(a._shape):
        raise IndexError("too many indices for tensor: tensor is %d-dimensional, but %d were indexed" % (len(a._shape), axis))
    plan.basic = tuple(basic)
    try:
        bshape = _broadcast_shapes(*[g[1]._shape for g in groups if isinstance(g[1], HipTensor)])
    except ValueError:
        raise IndexError("shape mismatch: indexing arrays could not be broadcast together with shapes %s"
                         % " ".join(str(g[1]._shape) for g in groups if isinstance(g[1], HipTensor)))
    plan.index = _fold_indices(groups, bshape)
    plan.merge = len(adv_axes)
    if adjacent:
        plan.axis = adv_axes[0]                   # the covered axes are neighbours: the index dimensions take their place
    else:
        # numpy: index arrays separated by a slice -> the index dimensions go FIRST.  A transposed view brings the covered axes
        # together at the front; the gather then runs along them as one axis
        nd = len(a._shape)
        plan.perm = tuple(adv_axes) + tuple(d for d in range(nd) if d not in adv_axes)
        plan.axis = 0
    return plan


def _plan_view(a, plan):
    """the basic view a plan's gather / assignment / scatter-add works on (axes permuted when the index arrays were apart)"""
    view = _idx_view(a, plan.basic)
    return view if plan.perm is None else view.transpose(*plan.perm)


def _take_extents(view_shape, plan):
    """(outer, axis_len, inner, pair_period, result shape) of a plan applied to a dense tensor of `view_shape`"""
    ax = plan.axis
    prod = lambda dims: int(np.prod(dims, dtype=np.int64)) if len(dims) else 1      # noqa: E731
    if plan.merge > 1:
        view_shape = view_shape[:ax] + (prod(view_shape[ax:ax + plan.merge]),) + view_shape[ax + plan.merge:]
    if plan.pair:
        n = view_shape[ax]
        return prod(view_shape[:ax]) * n, view_shape[ax + 1], prod(view_shape[ax + 2:]), n, view_shape[:ax] + (n,) + view_shape[ax + 2:]
    return prod(view_shape[:ax]), view_shape[ax], prod(view_shape[ax + 1:]), 0, view_shape[:ax] + plan.index._shape + view_shape[ax + 1:]


def _take(a, plan):
    src = _plan_view(a, plan).contiguous()
    outer, axis_len, inner, period, shape = _take_extents(src._shape, plan)
    out = HipTensor.empty(shape, dtype=a._dtype)
    _l.check(_l.lib().lg_take_axis(a._dtype.itemsize, src.ptr, outer, axis_len, inner, plan.index.ptr, plan.index._dtype.itemsize,
                                   plan.index.numel(), period, out.ptr))
    return out


def _put(a, plan, val):
    """a[idx] = val for a plan: in place when the basic view is dense, else through a dense copy of the view"""
    view = _plan_view(a, plan)
    dense = view if view.is_contiguous() else view.contiguous()
    outer, axis_len, inner, period, shape = _take_extents(dense._shape, plan)
    bits, vptr = 0, None
    if isinstance(val, np.ndarray) and val.ndim > 0:
        val = HipTensor.from_numpy(val.astype(a._dtype), requires_grad=False)
    if isinstance(val, HipTensor):
        assert val._dtype == a._dtype, "setitem: dtype mismatch (%s <- %s)" % (a._dtype, val._dtype)
        assert _broadcast_shapes(shape, val._shape) == shape, "setitem: value of shape %s does not broadcast to %s" % (val._shape, shape)
        if val._shape != shape or not val.is_contiguous():
            full = HipTensor.empty(shape, dtype=a._dtype, requires_grad=False)
            _l.check(_l.lib().lg_copy_strided(a._dtype.itemsize, len(shape), i64(shape), full.ptr, i64(full._strides), val.ptr,
                                              i64(_bstrides(val, shape))))
            val = full
        vptr = val.ptr
    else:
        bits = _value_bits(val, a._dtype)
    _l.check(_l.lib().lg_put_axis(a._dtype.itemsize, dense.ptr, outer, axis_len, inner, plan.index.ptr, plan.index._dtype.itemsize,
                                  plan.index.numel(), period, vptr, bits))
    if dense is not view:
        _l.check(_l.lib().lg_copy_strided(a._dtype.itemsize, len(view._shape), i64(view._shape), view.ptr, i64(view._strides),
                                          dense.ptr, i64(dense._strides)))


def _scatter_add(grad, plan, out_grad):
    """grad[idx] += out_grad for a plan (fp32); `grad` is a dense tensor of the indexed tensor's shape"""
    view = _plan_view(grad, plan)
    g = out_grad.contiguous()
    if view.is_contiguous():
        outer, axis_len, inner, period, shape = _take_extents(view._shape, plan)
        assert g._shape == shape
        _l.check(_l.lib().lg_scatter_add_axis_f32(view.ptr, outer, axis_len, inner, plan.index.ptr, plan.index._dtype.itemsize,
                                                  plan.index.numel(), period, g.ptr))
        return
    dense = HipTensor.zeros(view._shape, requires_grad=False)
    outer, axis_len, inner, period, shape = _take_extents(dense._shape, plan)
    _l.check(_l.lib().lg_scatter_add_axis_f32(dense.ptr, outer, axis_len, inner, plan.index.ptr, plan.index._dtype.itemsize,
                                              plan.index.numel(), period, g.ptr))
    _binary(_l.EW_ADD, view, dense, out=view)


def _idx_view(a, idx):
    """basic indexing (ints, slices, Ellipsis, None) as a strided view (opencl/ops.py:299-313, plus steps)"""
    idx = idx if isinstance(idx, tuple) else (idx,)
    if any(_is_advanced(i) for i in idx):
        raise NotImplementedError("integer-array indices go through _take_plan")
    n_real = _py.sum(1 for i in idx if i is not None and i is not Ellipsis)
    if Ellipsis in idx:
        k = idx.index(Ellipsis)
        idx = idx[:k] + (slice(None),) * (len(a._shape) - n_real) + idx[k + 1:]
    else:
        idx = idx + (slice(None),) * (len(a._shape) - n_real)
    shape, strides, offset, d = [], [], a._offset, 0
    for i in idx:
        if i is None:
            shape.append(1)
            strides.append(0)
            continue
        size, st = a._shape[d], a._strides[d]
        if isinstance(i, slice):
            start, stop, step = i.indices(size)
            n = len(range(start, stop, step))
            shape.append(n)
            strides.append(st * step)
            offset += start * st
        else:
            i = int(i)
            if i < -size or i >= size:
                raise IndexError("index %d is out of bounds for axis %d with size %d" % (i, d, size))
            offset += (i % size) * st
        d += 1
    return HipTensor(a.data, tuple(shape), tuple(strides), offset, a._dtype)


@HipTensor.register_op("__getitem__")
class getitem(Function):
    """ view + dense copy (opencl/ops.py:315-329); backward scatters into zeros (cpu/ops.py:242-246) """
    def forward(ctx, a, idx):
        if isinstance(idx, HipTensor) and a._dtype == _F32 and idx._dtype in (np.dtype(np.int32), np.dtype(np.int64)):
            # integer tensor index on the first axis = embedding lookup (examples/bert.py:19-21 does it on the CPU)
            ctx.save_for_backward(a._shape, idx)
            return _gather_rows(a, idx)
        plan = _take_plan(a, idx)
        if plan is not None:
            # integer-array index on one axis / the (range, labels) pair: gather forward, scatter-add backward (csrc/index.hip)
            ctx.save_for_backward(a._shape, plan)
            return _take(a, plan)
        ctx.save_for_backward(a._shape, idx)
        return _idx_view(a, idx).copy()

    def backward(ctx, out_grad):
        shape, idx = ctx.get_saved_tensors()
        if isinstance(idx, _TakePlan):
            _require_f32(out_grad)
            grad = HipTensor.zeros(shape, requires_grad=False)
            _scatter_add(grad, idx, out_grad)
            return grad
        if isinstance(idx, HipTensor):
            return _table_rows_grad(ctx._parents[0], shape, idx, out_grad)
        grad = HipTensor.zeros(shape, dtype=out_grad._dtype, requires_grad=False)
        grad[idx] = out_grad
        return grad


def _table_rows_grad(table, shape, idx, out_grad):
    """gradient of `table[idx]` (idx an integer tensor on the first axis): repeated ids accumulate.  A leaf table that already
    owns a gradient buffer (an embedding matrix after zero_grad) gets the rows added in place - no table-sized zero fill, no
    table-sized `grad +=` - and None is returned; otherwise the gradient tensor."""
    acc = table._grad_accumulator() if table.requires_grad else None
    if acc is not None and acc.is_contiguous() and acc._dtype == _F32:
        if table._consume_zero_pending():
            acc.fill(0)
        if GradGroup.usable_for(table) and idx.numel() <= 4096:
            ids_c, g_c = idx.contiguous(), out_grad.contiguous()        # (copies, if any, happen now, on the main chain)
            with GradGroup.issue(reads=(ids_c, g_c), writes=(acc,)):    # queued: leaves with the LayerNorm gradients
                _scatter_add_rows(shape, ids_c, g_c, into=acc)
        else:
            _scatter_add_rows(shape, idx, out_grad, into=acc)
        table._notify_grad_written()
        return None
    return _scatter_add_rows(shape, idx, out_grad)


def _tiled_ids(ids, shape):
    """the id tensor `ids` repeated along leading axes up to `shape`, as a dense tensor of its own.  Position ids are constants
    of a model, so the copy is kept with the id tensor's STORAGE (HipBuffer.derived) - every in-place writer into that storage
    (upload_, setitem, fill: tensor.flush_lazy_readers) drops it, and inside a hipGraph capture it is made anew each time, so
    that a replay tiles whatever the id buffer holds then - unless the model has declared the ids constant (`HipTensor.freeze()`)."""
    from .graph import HipGraph
    buf = ids.data
    key = (ids._offset, ids._shape, ids._strides, tuple(shape))
    keep = buf.frozen or not HipGraph.capturing          # (a frozen tensor cannot be refreshed: its copy may live in a graph)
    if keep and buf.derived is not None:
        hit = buf.derived.get(key)
        if hit is not None:
            return hit
    lead = len(shape) - len(ids._shape)
    tiled = HipTensor(buf, tuple(shape), (0,) * lead + tuple(ids._strides), ids._offset, ids._dtype, requires_grad=False).contiguous()
    if keep:
        if buf.derived is None:
            buf.derived = {}
        buf.derived[key] = tiled
    return tiled


# repeats of a shared id (position ids over a batch) up to which the scatter-add of the tiled ids stays in position order
# without atomics: csrc/tail_jobs.h kChunk
_SCATTER_ORDERED_REPEATS = 32


@HipTensor.register_op()
class embedding_sum(Function):
    """ (t0[ids0] + t1[ids1]) + t2[ids2] in one pass: the word, position and token-type lookups of a BERT embedding layer and
    their two additions (reference examples/bert.py:36-40; there on the CPU).  The id tensors go in by keyword (they take no
    gradient); ids0 has the full shape, ids1 / ids2 may lack leading axes (position ids shared by the batch).  Same bits as the
    three `table[ids]` and two `+` of the tape; the backward is theirs too: rows added into each table's gradient. """
    def forward(ctx, t0, t1, t2, ids0=None, ids1=None, ids2=None):
        _require_f32(t0, t1, t2)
        ids = (ids0, ids1, ids2)
        tables = tuple(t.contiguous() for t in (t0, t1, t2))
        assert all(isinstance(i, HipTensor) and i._dtype == ids0._dtype for i in ids) and ids0._dtype in (np.dtype(np.int32), np.dtype(np.int64)), \
            "embedding_sum: the three id tensors must be int32 or int64 tensors of one dtype"
        assert all(len(t._shape) == 2 and t._shape[1] == t0._shape[1] for t in tables), "embedding_sum: tables of one row length"
        for i in ids[1:]:
            assert len(i._shape) <= len(ids0._shape) and ids0._shape[len(ids0._shape) - len(i._shape):] == i._shape, \
                "embedding_sum: ids of shape %s do not broadcast against %s by leading axes" % (i._shape, ids0._shape)
        ids = tuple(i.contiguous() for i in ids)
        out = HipTensor.empty(ids0._shape + (t0._shape[1],))
        args = []
        for t, i in zip(tables, ids):
            args += [t.ptr, i.ptr, i.numel(), t._shape[0]]
        _l.check(_l.lib().lg_gather_sum3_rows_f32(*args, ids0._dtype.itemsize, out.ptr, ids0.numel(), t0._shape[1]))
        ctx.save_for_backward(ids, tuple(t._shape for t in tables))
        return out

    def backward(ctx, out_grad):
        ids, shapes = ctx.get_saved_tensors()
        grads = []
        for table, shape, i in zip(ctx._parents[:3], shapes, ids):
            if not table.requires_grad:
                grads.append(None)
                continue
            g = out_grad
            if i._shape != ids[0]._shape:
                lead = len(ids[0]._shape) - len(i._shape)
                repeats = 1
                for n in ids[0]._shape[:lead]:
                    repeats *= n
                if repeats <= _SCATTER_ORDERED_REPEATS:
                    # ids shared by leading axes (position ids of a batch): every output row is scattered with its own copy of
                    # the id - the scatter-add sums the repeats in position order, no separate sum over the batch in front of it
                    i = _tiled_ids(i, ids[0]._shape)
                else:
                    # more repeats than one ordered chunk of the scatter kernel holds: every shared id would be a "hot" row summed
                    # with float atomics (not reproducible, not the reference's order).  The tape's own form instead: the sum over
                    # the leading axes first (func.py:50-56's un-broadcast), then one ordered scatter
                    g = _reduce(_l.RED_SUM, out_grad.contiguous(), tuple(range(lead)), False)
            grads.append(_table_rows_grad(table, shape, i, g))
        return tuple(grads)


@HipTensor.register_op("__setitem__")
class setitem(Function):
    """ strided copy / fill into the indexed view (opencl/ops.py:331-340) """
    def forward(ctx, a, idx, val):
        flush_lazy_readers(a)
        plan = _take_plan(a, idx)
        if plan is not None:
            _put(a, plan, val)
            return a
        view = _idx_view(a, idx)
        if isinstance(val, np.ndarray) and val.ndim > 0:
            val = HipTensor.from_numpy(val.astype(a._dtype), requires_grad=False)
        if isinstance(val, HipTensor):
            assert val._dtype == a._dtype, "setitem: dtype mismatch (%s <- %s)" % (a._dtype, val._dtype)
            assert _broadcast_shapes(view._shape, val._shape) == view._shape, \
                "setitem: value of shape %s does not broadcast to %s" % (val._shape, view._shape)
            _l.check(_l.lib().lg_copy_strided(a._dtype.itemsize, len(view._shape), i64(view._shape), view.ptr,
                                              i64(view._strides), val.ptr, i64(_bstrides(val, view._shape))))
        else:
            view.fill(val)
        return a


""" Reductions """


def _norm_axes(nd, axis):
    if axis is None:
        return tuple(range(nd))
    axes = axis if isinstance(axis, (tuple, list)) else (axis,)
    axes = tuple(sorted(a % nd for a in axes)) if nd > 0 else ()
    assert len(set(axes)) == len(axes), "duplicate value in 'axis'"
    return axes


def _reduce_into(acc, x, axes, overwrite=False):
    """acc += x.sum(axes) (or acc = ... after a lazy zero_grad): the reduction's final pass adds into an existing dense
    buffer (lg_reduce_acc)"""
    _require_f32(x, acc)
    mask = 0
    for a in axes:
        mask |= 1 << a
    kept = tuple(s for i, s in enumerate(x._shape) if i not in axes)
    assert acc._shape == kept and acc.is_contiguous()
    flush_lazy_readers(acc)
    _l.check(_l.lib().lg_reduce_acc(_l.RED_SUM, len(x._shape), i64(x._shape), x.ptr, i64(x._strides), mask, acc.ptr,
                                    0 if overwrite else 1))


def _reduce(op, x, axes, keepdims):
    mask = 0
    for a in axes:
        mask |= 1 << a
    kept = tuple(s for i, s in enumerate(x._shape) if i not in axes)
    if x._dtype != _F32:
        code = _TYPED.get(x._dtype)
        if code is None:
            raise TypeError("HipTensor reductions: dtype %s (float32, float64, int16, int32, int64)" % x._dtype)
        # numpy: the sum of int16 / int32 values is formed - and returned - in int64; max / min keep the dtype
        out = HipTensor.empty(kept, dtype=np.int64 if (op == _l.RED_SUM and x._dtype != _F64) else x._dtype)
        _l.check(_l.lib().lg_reduce_typed(op, code, len(x._shape), i64(x._shape), x.ptr, i64(x._strides), mask, out.ptr))
    else:
        out = HipTensor.empty(kept)
        _l.check(_l.lib().lg_reduce(op, len(x._shape), i64(x._shape), x.ptr, i64(x._strides), mask, out.ptr))
    if keepdims:
        full = tuple(1 if i in axes else s for i, s in enumerate(x._shape))
        out = HipTensor(out.data, full, None, out._offset, out._dtype)
    return out


def _keepdims_view(t, in_shape, axes, keepdims):
    """view of a reduced tensor with the reduced axes re-inserted as size-1 dims"""
    if keepdims:
        return t
    full = tuple(1 if i in axes else s for i, s in enumerate(in_shape))
    it = iter(t._strides)
    strides = tuple(0 if i in axes else next(it) for i in range(len(in_shape)))
    return HipTensor(t.data, full, strides, t._offset, t._dtype)


@HipTensor.register_op()
class sum(Function):
    """ forward cpu/ops.py:288-293; backward = zero-stride broadcast VIEW of the gradient, no kernel
    (the reference's only sum.backward: opencl/ops.py:353-368) """
    def forward(ctx, x, axis=None, keepdims=False):
        axes = _norm_axes(len(x._shape), axis)
        ctx.save_for_backward(x._shape, axes, keepdims)
        return _reduce(_l.RED_SUM, x, axes, keepdims)

    def backward(ctx, out_grad):
        shape, axes, keepdims = ctx.get_saved_tensors()
        g = _keepdims_view(out_grad, shape, axes, keepdims)
        return HipTensor(g.data, shape, _bstrides(g, shape), g._offset, g._dtype)


def _extremum(name, red_op, cite):
    class Op(Function):
        def forward(ctx, x, axis=None, keepdims=False):
            axes = _norm_axes(len(x._shape), axis)
            val = _reduce(red_op, x, axes, True)
            ctx.save_for_backward(x, _saved_output(val), axes, keepdims)
            if keepdims:
                return val
            return HipTensor(val.data, tuple(s for i, s in enumerate(x._shape) if i not in axes), None, val._offset, val._dtype)

        def backward(ctx, out_grad):
            x, val, axes, keepdims = ctx.get_saved_tensors()
            g = _keepdims_view(out_grad, x._shape, axes, keepdims)
            _require_f32(out_grad)
            return _ew(_l.EW_MAX_BWD, x._shape, [x, val, g])
    Op.__name__ = Op.__qualname__ = name
    Op.__doc__ = cite
    return Op


max = HipTensor.register_op("max", _extremum("max", _l.RED_MAX, "every tied maximum receives the gradient: g * (x == max) (cpu/ops.py:260-272)"))
min = HipTensor.register_op("min", _extremum("min", _l.RED_MIN, "cpu/ops.py:274-286"))


""" Convolution (CNN example, SURVEY.md §8f row 4 tail): window VIEW -> one gather copy -> MFMA GEMM """


def _conv_strides(strides, n):
    if isinstance(strides, int):
        return (1,) + (strides,) * (n - 1) if n > 1 else (strides,)
    strides = tuple(strides)
    return (1,) + strides if len(strides) == n - 1 else strides


@HipTensor.register_op()
class conv(Function):
    """ valid N-d cross-correlation t (..., C, d1..dm) * kernel (out_c, C, k1..km) -> (..., out_c, o1..om) with the
    semantics of cpu/ops.py:298-356.  All windows are ONE strided (overlapping) view of the input - no im2col loop -
    gathered once into the window matrix, which the SGEMM kernel multiplies with the flattened kernel.  backward:
    two GEMMs, then one strided in-place add per spatial kernel offset (slabs of one offset never overlap).  The
    reference's OpenCL backend has a direct kernel and no backward (opencl/ops.py:403-408). """
    def forward(ctx, t, kernel, strides=1):
        _require_f32(t, kernel)
        n = len(kernel._shape) - 1
        st = _conv_strides(strides, n)
        kshape = kernel._shape[1:]
        assert len(t._shape) >= n and len(st) == n and t._shape[-n] == kshape[0], \
            "conv: input %s does not match kernel %s" % (t._shape, kernel._shape)
        out_pos = tuple((d - k) // s + 1 for d, k, s in zip(t._shape[-n:], kshape, st))
        # window view without the channel-position axis (its extent is 1): (..., o1..om, C, k1..km)
        wshape = t._shape[:-n] + out_pos[1:] + kshape
        wstrides = t._strides[:-n] + tuple(a * s for a, s in zip(t._strides[-n + 1:], st[1:])) + t._strides[-n:]
        assert len(wshape) <= 8, "conv: window view needs %d dims (max 8)" % len(wshape)
        win = HipTensor(t.data, wshape, wstrides, t._offset, t._dtype)
        ncol = 1
        for k in kshape:
            ncol *= k
        cols = win.contiguous().reshape(-1, ncol)
        w2 = kernel.reshape(kernel._shape[0], ncol)
        y = _gemm(cols, _swap_last(w2))                                           # (P, out_c)
        ctx.save_for_backward(cols, w2, t._shape, kernel._shape, st, out_pos)
        lead = len(t._shape) - n
        y = y.reshape(*t._shape[:lead], *out_pos[1:], kernel._shape[0])
        perm = tuple(range(lead)) + (len(y._shape) - 1,) + tuple(range(lead, len(y._shape) - 1))
        return y.transpose(*perm).contiguous()

    def backward(ctx, out_grad):
        cols, w2, in_shape, k_shape, st, out_pos = ctx.get_saved_tensors()
        n = len(k_shape) - 1
        lead = len(in_shape) - n
        m = len(out_grad._shape)
        perm = tuple(range(lead)) + tuple(range(lead + 1, m)) + (lead,)                # out_c last
        g2 = out_grad.transpose(*perm).reshape(-1, k_shape[0])
        dw = _gemm(_swap_last(g2), cols).reshape(*k_shape)
        dwin = _gemm(g2, w2).reshape(*in_shape[:lead], *out_pos[1:], *k_shape[1:])   # (..., o1..om, C, k1..km)
        dx = HipTensor.zeros(in_shape, requires_grad=False)
        nsp = n - 1                                                                     # spatial axes
        for off in np.ndindex(*k_shape[2:]):
            src = _idx_view(dwin, (Ellipsis,) + off)                                   # (..., o1..om, C)
            ms = len(src._shape)
            src = src.transpose(*(tuple(range(lead)) + (ms - 1,) + tuple(range(lead, ms - 1))))   # (..., C, o1..om)
            dst = _idx_view(dx, (Ellipsis,) + tuple(slice(o, o + s * p, s) for o, s, p in zip(off, st[1:], out_pos[1:])))
            _binary(_l.EW_ADD, dst, src, out=dst)
        assert nsp == len(k_shape) - 2
        return dx, dw


""" Fused forms used by nn / optim / loss (SURVEY.md §8f row 1) """


@HipTensor.register_op()
class linear(Function):
    """ nn.Linear in one tape node: `x @ W.T(1, 0) + b` (nn.py:90-96) with the bias added in the GEMM epilogue.
    Same values as the three-op form (transpose view, dot, broadcast add): the product is rounded to fp32 before the
    bias is added.  backward: dx = g @ W, dW = g^T @ x (dense, in W's own layout), db = column sums of g - what
    dot.backward + transpose.backward + the un-broadcast of func.py:50-56 produce. """
    def forward(ctx, x, weight, bias=None, residual=None):
        ctx.save_for_backward(x, weight, bias is not None)
        if residual is not None:
            # `dense(h) + h_in` (reference examples/bert.py:101, :117): the residual is added in the GEMM's epilogue - the
            # values of the separate add, (x @ W^T + b) + residual, without its pass; its gradient is out_grad itself
            assert residual._shape == x._shape[:-1] + (weight._shape[0],), "linear: residual %s for an output of %s" % (
                residual._shape, x._shape[:-1] + (weight._shape[0],))
            _require_f32(x, weight, residual)
            if x.numel() > 0:
                return _gemm(x, _swap_last(weight), bias=bias, addend=residual.contiguous())
            return _binary(_l.EW_ADD, _gemm(x, _swap_last(weight), bias=bias), residual)
        pre = _lazy_relu_input(x)
        src = pre if pre is not None else x
        if _head_eligible(src, weight, bias):
            # a skinny output layer: wait and see whether the loss wants it - loss.mse then computes both in one launch
            out = HipTensor(None, (src._shape[0], weight._shape[0]), None, 0, _F32)
            out._lazy_source = ("head", src, pre is not None, weight, bias)
            out._watch_sources(src, weight, bias)
            return out
        if pre is not None and x._shape[0] > 0:
            # x = relu(pre) that nobody has looked at yet: the GEMM reads pre and applies the relu while staging it
            return _gemm_fused(pre, _swap_last(weight), bias=bias, relu_a=True)[0]
        if bias is None and pre is None and x._dtype == _F32 and weight._dtype == _F32 and len(x._shape) >= 1 \
                and x._shape[-1] > 0 and x.numel() // x._shape[-1] * weight._shape[0] >= _LAZY_LINEAR_MIN_OUT:
            # a large bias-free product (BERT's decoder onto the vocabulary): wait and see whether a bias row is added to it
            # next (`decoder(h) + bias`, reference bert.py:227) - HipTensor.add then runs ONE GEMM with the bias in
            # its epilogue instead of a second pass over the (rows, vocabulary) result
            out = HipTensor(None, x._shape[:-1] + (weight._shape[0],), None, 0, _F32)
            out._lazy_source = ("linear", x, weight)
            out._watch_sources(x, weight)
            return out
        return _gemm(x, _swap_last(weight), bias=bias)

    def backward(ctx, out_grad):
        grads = linear._backward(ctx, out_grad)
        if len(ctx._parents) > 3 and ctx._parents[3] is not None:
            # (x, weight, bias | None, residual): the residual's gradient is the output's
            grads = tuple(grads) + (None,) * (3 - len(grads)) + (out_grad,)
        return grads

    def _backward(ctx, out_grad):
        x, weight, has_bias = ctx.get_saved_tensors()
        bias = ctx._parents[2] if has_bias else None
        out_f = weight._shape[0]
        g2 = _rows(out_grad, out_f)
        g2._unfinished_loss = out_grad._unfinished_loss        # a view of the same `err` (see head_mse_forward)
        g2._head_grad_ahead = out_grad._head_grad_ahead
        pre = _lazy_relu_input(x) if g2._shape[0] > 0 else None
        if pre is not None:
            return linear._backward_through_lazy_relu(x, pre, weight, bias, g2)
        if (weight.requires_grad and x.requires_grad and len(x._shape) == 2 and _head_eligible(x, weight, bias) and g2.is_contiguous()):
            return _head_backward(x, x, False, weight, bias, g2)
        x2 = _rows(x, x._shape[-1])
        # leaf operands that already own a gradient buffer (parameters after zero_grad, a re-used input) get their
        # gradient ADDED in place by the producing kernel (GEMM with beta = 1 / reduction with accumulate) and None
        # is reported for them - tensor.py:118's `grad += g` without the temporary and the extra pass
        want_db = has_bias and bias.requires_grad
        acc_w = weight._grad_accumulator() if weight.requires_grad else None
        acc_w = acc_w if (acc_w is not None and acc_w.is_contiguous()) else None
        acc_b = bias._grad_accumulator() if want_db else None
        acc_b = acc_b if (acc_b is not None and acc_b.is_contiguous()) else None
        rows = g2._shape[0]
        in_place = rows > 0 and x.requires_grad and acc_w is not None and (not want_db or acc_b is not None)
        if in_place and GradGroup.usable_for(weight, bias):
            # deep tape: dW (+ db) are QUEUED inside the library and launched with all the other weight gradients when the
            # pass ends; the main chain carries on with dx, the only result the rest of the backward pass waits for
            with GradGroup.issue(reads=(g2, x2), writes=(acc_w, acc_b)):
                linear._weight_products(x2, weight, bias, want_db, g2, acc_w, acc_b)
            return linear._input_product(x, weight, g2) + ((None,) if not has_bias else (None, None))
        # dW (+ db) and dx are independent products: when both are wanted they go out as ONE launch (lg_gemm_pair_*) - unless a
        # data-parallel exchange hangs on the weight gradient's kernel being enqueued the moment it is reported written
        paired = (weight.requires_grad and x.requires_grad and rows > 0 and weight._grad_written_hook is None
                  and (bias is None or bias._grad_written_hook is None))
        kept = None
        if paired:
            if HeldPair.resume():
                # the skinny output layer behind this one left its weight gradient in an open bracket (_head_backward): the two
                # products below join it - three products and the loss in one launch
                kept = HeldPair.keep
                HeldPair.done()
            else:
                _l.check(_l.lib().lg_gemm_pair_begin())
        try:
            dw, db = linear._weight_products(x2, weight, bias, want_db, g2, acc_w, acc_b)
            dx, = linear._input_product(x, weight, g2)
        finally:
            if paired:
                _l.check(_l.lib().lg_gemm_pair_end())
        del kept
        return (dx, dw, db) if has_bias else (dx, dw)

    @staticmethod
    def _weight_products(x2, weight, bias, want_db, g2, acc_w, acc_b):
        """(dW, db) = (g^T @ x, column sums of g): each added into its accumulator when there is one (None is returned for
        it then and the parameter's hook is told), else returned as a fresh tensor; None for a gradient nobody wants"""
        dw = db = None
        rows = g2._shape[0]
        if weight.requires_grad and want_db and rows > 0 and _rowsum_column_is_cheap(weight._shape[0], weight._shape[1]):
            # dW and db from one launch: db = column sums of g = row sums of g^T, a virtual extra column of the product
            out_w, out_b = _gemm_rowsum(_swap_last(g2), x2, accumulate_into=acc_w,
                                        overwrite=acc_w is not None and weight._consume_zero_pending(),
                                        rowsum_into=acc_b, rowsum_overwrite=acc_b is not None and bias._consume_zero_pending())
            dw = out_w if acc_w is None else None
            db = out_b if acc_b is None else None
            want_db = False
        elif weight.requires_grad:
            if acc_w is not None:
                _gemm(_swap_last(g2), x2, accumulate_into=acc_w, overwrite=weight._consume_zero_pending())
            else:
                dw = _gemm(_swap_last(g2), x2)
        if want_db:
            if acc_b is not None and GradGroup.issuing and g2.is_contiguous() and g2._dtype == _F32:
                # queued with the weight gradients: the column sums are computed by extra workgroups of the group's launch
                flush_lazy_readers(acc_b)
                _l.check(_l.lib().lg_gemm_group_colsum_f32(g2.ptr, g2._shape[1], g2._shape[0], g2._shape[1], acc_b.ptr,
                                                            0 if bias._consume_zero_pending() else 1))
            elif acc_b is not None:
                _reduce_into(acc_b, g2, (0,), overwrite=bias._consume_zero_pending())
            else:
                db = _reduce(_l.RED_SUM, g2, (0,), False)
        if weight.requires_grad and acc_w is not None:
            weight._notify_grad_written()
        if bias is not None and bias.requires_grad and acc_b is not None:
            bias._notify_grad_written()
        return dw, db

    @staticmethod
    def _input_product(x, weight, g2):
        """(dx,) = (g @ W,), or (None,) after adding it into x's own gradient buffer / when x wants none"""
        if not x.requires_grad:
            return (None,)
        acc = x._grad_accumulator()
        if acc is not None and acc.is_contiguous() and len(x._shape) == 2:
            _gemm(g2, weight, accumulate_into=acc, overwrite=x._consume_zero_pending())
            (x._view_of_leaf if (x._view_of_leaf is not None and x._grad is None) else x)._notify_grad_written()
            return (None,)
        have = x._grad if (x._ctx is not None and x._view_of_leaf is None) else None
        if (have is not None and have.__class__ is HipTensor and have._shape == x._shape and have._dtype == _F32
                and have.is_contiguous() and g2._shape[0] > 0 and weight._shape[0] > 0):
            # an intermediate that already holds a contribution (the residual branch, a sibling projection): what add_grad
            # would do - `grad = grad + dx`, then `grad += dx` (tensor.py:111-118) - happens in this GEMM's epilogue
            flat = HipTensor(have.data, (g2._shape[0], weight._shape[1]), None, have._offset, have._dtype)
            if x._grad_shared:
                x._grad = _gemm(g2, weight, addend=flat).reshape(*x._shape)      # the shared tensor stays untouched
                x._grad._requires_grad = False
                x._grad_shared = False
            else:
                _gemm(g2, weight, accumulate_into=flat)
            return (None,)
        return (_gemm(g2, weight).reshape(*x._shape),)


@HipTensor.register_op()
class feed_forward(Function):
    """ dense2(gelu(dense1(x))) + residual in one tape node - the position-wise block of a transformer layer (reference
    examples/bert.py:104-118: Linear, gelu, Linear, `+ hidden`): the gelu rides in the first product's epilogue, the residual in
    the second's, and in the backward the gelu derivative in the epilogue of the product that makes the gradient it scales.
    Four launches instead of six, the values of the separate ops bit for bit.  Weights as nn.Linear holds them ((out, in));
    the weight / bias gradients take the routes of `linear` (queued in a deep backward pass, else with dx). """
    def forward(ctx, x, w1, b1, w2, b2, residual=None):
        _require_f32(x, w1, b1, w2, b2)
        assert b1 is not None and b2 is not None and x.numel() > 0
        x2 = x.reshape(-1, x._shape[-1])
        pre, act = _gemm_act(x2, _swap_last(w1), _l.ACT_GELU, bias=b1.contiguous())
        if residual is not None:
            assert residual._shape == x._shape[:-1] + (w2._shape[0],)
            res2 = residual.contiguous()
            out = _gemm(act, _swap_last(w2), bias=b2, addend=HipTensor(res2.data, (x2._shape[0], w2._shape[0]), None, res2._offset, res2._dtype))
        else:
            out = _gemm(act, _swap_last(w2), bias=b2)
        ctx.save_for_backward(x, pre, act)
        return out.reshape(*(x._shape[:-1] + (w2._shape[0],)))

    def backward(ctx, out_grad):
        x, pre, act = ctx.get_saved_tensors()
        _, w1, b1, w2, b2 = ctx._parents[:5]
        g2 = out_grad.reshape(-1, w2._shape[0]).contiguous()
        x2 = x.reshape(-1, x._shape[-1])
        grads = {}

        def weight_side(name_w, name_b, inp, weight, bias, g):
            want_db = bias.requires_grad
            acc_w = weight._grad_accumulator() if weight.requires_grad else None
            acc_w = acc_w if (acc_w is not None and acc_w.is_contiguous()) else None
            acc_b = bias._grad_accumulator() if want_db else None
            acc_b = acc_b if (acc_b is not None and acc_b.is_contiguous()) else None
            if acc_w is not None and (not want_db or acc_b is not None) and GradGroup.usable_for(weight, bias):
                with GradGroup.issue(reads=(g, inp), writes=(acc_w, acc_b)):
                    grads[name_w], grads[name_b] = linear._weight_products(inp, weight, bias, want_db, g, acc_w, acc_b)
            else:
                grads[name_w], grads[name_b] = linear._weight_products(inp, weight, bias, want_db, g, acc_w, acc_b)
        weight_side("w2", "b2", act, w2, b2, g2)
        dpre, _ = _gemm_act(g2, w2, _l.ACT_GELU_BWD, aux=pre)        # (g @ W2) * gelu'(pre)
        weight_side("w1", "b1", x2, w1, b1, dpre)
        residual = ctx._parents[5] if len(ctx._parents) > 5 else None
        res_grad = out_grad
        if residual is x and x.requires_grad and x._ctx is not None and x._view_of_leaf is None:
            # the block's input is its residual: out_grad goes to it first, and the product below adds dx to it in its
            # epilogue (linear._input_product) instead of a separate `grad + dx` pass
            x.add_grad(out_grad)
            res_grad = None
        dx, = linear._input_product(x, w1, dpre)
        out = (dx, grads["w1"], grads["b1"], grads["w2"], grads["b2"])
        if residual is not None:
            out = out + (res_grad,)
        return out


def _linear_backward_through_lazy_relu(x, pre, weight, bias, g2):
    """linear.backward when the saved input x = relu(pre) was never made: dW (+ db) read pre with the relu applied on
    the fly.  dx is the plain g @ W: the relu output's own gradient stays what the tape says it is (its (pre >= 0)
    factor is relu.backward's job).  Same values as with a materialised x."""
    want_db = bias is not None and bias.requires_grad
    if weight.requires_grad and _head_eligible(pre, weight, bias) and g2.is_contiguous():
        return _head_backward(x, pre, True, weight, bias, g2)
    # dW (+ db) and dx as ONE launch, like linear._backward's `paired` (a hidden layer between two others: Linear -> relu -> THIS
    # -> relu -> ...), and with them whatever a skinny output layer behind this one left in the bracket (_head_backward_riding)
    paired = (weight.requires_grad and x.requires_grad and g2._shape[0] > 0 and weight._grad_written_hook is None
              and (bias is None or bias._grad_written_hook is None))
    kept = None
    if paired:
        if HeldPair.resume():
            kept = HeldPair.keep
            HeldPair.done()
        else:
            _l.check(_l.lib().lg_gemm_pair_begin())
    try:
        return _linear_backward_through_lazy_relu_products(x, pre, weight, bias, g2, want_db)
    finally:
        if paired:
            _l.check(_l.lib().lg_gemm_pair_end())
        del kept


def _linear_backward_through_lazy_relu_products(x, pre, weight, bias, g2, want_db):
    dw = dx = db = None
    if weight.requires_grad:
        acc_w = weight._grad_accumulator()
        acc_w = acc_w if (acc_w is not None and acc_w.is_contiguous()) else None
        acc_b = bias._grad_accumulator() if want_db else None
        acc_b = acc_b if (acc_b is not None and acc_b.is_contiguous()) else None
        out_w, out_b = _gemm_fused(_swap_last(g2), pre, relu_b=True, accumulate_into=acc_w,
                                   overwrite=acc_w is not None and weight._consume_zero_pending(),
                                   want_rowsum=want_db, rowsum_into=acc_b,
                                   rowsum_overwrite=acc_b is not None and bias._consume_zero_pending())
        dw = out_w if acc_w is None else None
        if acc_w is not None:
            weight._notify_grad_written()
        if want_db:
            db = out_b if acc_b is None else None
            if acc_b is not None:
                bias._notify_grad_written()
            want_db = False
    if x.requires_grad:
        dx = _gemm(g2, weight)
    if bias is None:
        return dx, dw
    if want_db:
        acc = bias._grad_accumulator()
        if acc is not None and acc.is_contiguous():
            _reduce_into(acc, g2, (0,), overwrite=bias._consume_zero_pending())
            bias._notify_grad_written()
        else:
            db = _reduce(_l.RED_SUM, g2, (0,), False)
    return dx, dw, db


linear._backward_through_lazy_relu = staticmethod(_linear_backward_through_lazy_relu)


_LAZY_LINEAR_MIN_OUT = 1 << 20      # elements of the product from which a bias-free Linear waits for its consumer


_tape_add = HipTensor.add


def _add_folding_bias(self, other):
    """`a + b`.  Peephole: a = x @ W^T of a bias-free Linear that nobody has looked at yet (lazy, see linear.forward) and b one
    value per output feature -> a single `linear` node over (x, W, b), the bias added in the GEMM epilogue.  Same values (the
    product is rounded to fp32 before the bias is added) and the same gradients for x, W and b as the two-node form; should `a`
    be used elsewhere too, its own node still carries that use."""
    src = self._lazy_source
    if (src is not None and self._data is None and src[0] == "linear" and other.__class__ is HipTensor
            and other._shape == self._shape[-1:] and other._dtype == _F32 and other.is_contiguous()):
        return linear(src[1], src[2], other)
    return _tape_add(self, other)


_add_folding_bias.__name__ = "add"
HipTensor.add = _add_folding_bias


def _rowsum_column_is_cheap(out_features: int, in_features: int) -> bool:
    """db rides in the dW GEMM as a virtual extra column of the (out_features, in_features) product.  That is free when the
    last 64-wide tile column is partial anyway, and harmless while the whole grid fits the chip in one round; a full extra tile
    column on a throughput-bound product (BERT's decoder: 477 x 2 -> 477 x 3 tiles) costs more than a reduction of its own."""
    tiles_m, tiles_n = -(-out_features // 64), -(-(in_features + 1) // 64)
    return in_features % 64 != 0 or tiles_m * tiles_n <= 256


def _head_eligible(x, weight, bias) -> bool:
    """a Linear small enough for csrc/head.hip: dense 2-D fp32 input (or pre-activation) of at most 4096 rows (the weight
    gradient is reduced over ALL rows by one workgroup per 8 columns: beyond that the MFMA GEMM is the better tool),
    <= 16 output features, the alignment its float4 loads need and a weight matrix that fits the 64 KiB LDS stage"""
    if len(x._shape) != 2 or len(weight._shape) != 2 or x._dtype != _F32 or weight._dtype != _F32:
        return False
    rows, hidden = x._shape
    outs = weight._shape[0]
    if rows < 1 or rows > 4096 or not 1 <= outs <= 16 or hidden < 4 or hidden % 4 or outs * hidden * 4 > 65536 or weight._shape[1] != hidden:
        return False
    if not (x.is_contiguous() and weight.is_contiguous()) or (bias is not None and not (bias.is_contiguous() and bias._dtype == _F32)):
        return False
    return x.ptr % 16 == 0 and weight.ptr % 16 == 0


def head_mse_forward(y, y_hat):
    """loss.mse of a still-lazy skinny output layer `y`: (loss, err) AND y itself from one launch (lg_head_fwd_f32).
    Returns None when the fused form does not apply (the caller then materialises y the plain way)."""
    _, x, relu, weight, bias = y._lazy_source
    # (linear.forward made `y` lazy only after _head_eligible said yes; its operands have not changed shape or place since)
    if not isinstance(y_hat, HipTensor) or y_hat._shape != y._shape or y_hat._dtype != _F32:
        return None
    y_hat = y_hat.contiguous()
    rows, hidden = x._shape
    outs = weight._shape[0]
    out, err, row_loss = HipTensor.empty(y._shape, requires_grad=False), HipTensor.empty(y._shape), HipTensor.empty((rows,), requires_grad=False)
    # The tape is recording (y has a node) and relu(x) @ W^T is what y is: the launch also writes relu.backward's result for the
    # case that backward() starts at this loss - dx = err @ W and g_pre = dx * (x >= 0), 10 multiply-adds per element next to a row that is in
    # the cache anyway, instead of a launch of its own in the backward pass (head_bwd's tile workgroups: 7.4 us at 1024 x 512).
    # Whether the backward pass may use it is decided there (_head_grad_ahead_of).
    dx = gpre = None
    if relu and _HEAD_GRAD_AHEAD and y._ctx is not None and x._requires_grad and weight._requires_grad and hidden % 4 == 0:
        dx, gpre = HipTensor.empty((rows, hidden)), HipTensor.empty((rows, hidden), requires_grad=False)
    _l.check(_l.lib().lg_head_fwd_grad_f32(x.ptr, hidden, 1 if relu else 0, weight.ptr, bias.ptr if bias is not None else None, y_hat.ptr,
                                           out.ptr, err.ptr, row_loss.ptr, dx.ptr if dx is not None else None,
                                           gpre.ptr if gpre is not None else None, rows, hidden, outs))
    y._data, y._offset, y._byte_offset, y._lazy_source = out._data, out._offset, out._byte_offset, None           # y is real now
    if gpre is not None:
        token = object()
        for buf in (x._data, weight._data, err._data):
            if buf.derived is None:
                buf.derived = {}
            buf.derived["head_grad"] = token
        err._head_grad_ahead = (x, weight, dx, gpre, token)
    # the scalar loss stays lazy: the backward launch of this head finishes it (a cross-workgroup sum inside the forward
    # launch would cost 4 us); whoever reads it before that pays one small launch
    loss = HipTensor(None, (), None, 0, _F32)
    loss._lazy_source = ("mse_rows", row_loss, rows * outs)
    err._unfinished_loss = weakref.ref(loss)
    return loss, err


def _head_backward(x, src, relu, weight, bias, g2):
    """linear.backward of a skinny output layer in ONE launch (lg_head_bwd_f32): dx, dW, db - and, when the layer's
    input is a lazy relu(src), that relu's backward result too, handed to relu.backward through `_relu_bwd_done`."""
    rows, hidden = src._shape
    outs = weight._shape[0]
    want_db = bias is not None and bias.requires_grad
    acc_w = weight._grad_accumulator()
    acc_w = acc_w if (acc_w is not None and acc_w.is_contiguous()) else None
    acc_b = bias._grad_accumulator() if want_db else None
    acc_b = acc_b if (acc_b is not None and acc_b.is_contiguous()) else None
    if _head_weight_gradient_can_ride(x, src, relu, weight, bias, acc_w, acc_b, want_db):
        return _head_backward_riding(x, src, relu, weight, bias, g2, acc_w, acc_b, want_db)
    for t in (acc_w, acc_b):
        if t is not None:
            flush_lazy_readers(t)
    dw = acc_w if acc_w is not None else HipTensor.empty(weight._shape)
    db = (acc_b if acc_b is not None else HipTensor.empty((outs,))) if want_db else None
    ahead = _head_grad_ahead_of(g2, src, weight) if (relu and x.requires_grad) else None
    dx = HipTensor.empty((rows, hidden)) if (x.requires_grad and ahead is None) else None
    gpre = HipTensor.empty((rows, hidden)) if (relu and dx is not None) else None
    # g2 is the `err` of a fused head + mse forward whose scalar loss nobody has looked at yet: finish it in this launch
    loss = g2._unfinished_loss() if g2._unfinished_loss is not None else None
    row_loss = loss_out = None
    if loss is not None and loss._data is None and loss._lazy_source is not None and loss._lazy_source[0] == "mse_rows" \
            and loss._lazy_source[1]._shape == (rows,) and loss._lazy_source[2] == rows * outs:
        row_loss, loss_out = loss._lazy_source[1], HipTensor.empty((), requires_grad=False)
    _l.check(_l.lib().lg_head_bwd_f32(
        src.ptr, hidden, 1 if relu else 0, g2.ptr, weight.ptr,
        dx.ptr if dx is not None else None, gpre.ptr if gpre is not None else None,
        dw.ptr, 1 if (acc_w is not None and not weight._consume_zero_pending()) else 0,
        db.ptr if db is not None else None, 1 if (acc_b is not None and not bias._consume_zero_pending()) else 0,
        rows, hidden, outs, row_loss.ptr if row_loss is not None else None, loss_out.ptr if loss_out is not None else None))
    if loss_out is not None:
        loss._data, loss._offset, loss._byte_offset, loss._lazy_source = loss_out._data, loss_out._offset, loss_out._byte_offset, None
        g2._unfinished_loss = None
    if acc_w is not None:
        weight._notify_grad_written()
    if acc_b is not None:
        bias._notify_grad_written()
    if ahead is not None:
        dx, gpre = ahead           # written by the forward launch (head_mse_forward): this launch had the weight gradients only
    if gpre is not None:
        dx._relu_bwd_done = (src, gpre)
    if dx is not None and len(x._shape) != 2:
        dx = dx.reshape(*x._shape)
    if bias is None:
        return dx, (None if acc_w is not None else dw)
    return dx, (None if acc_w is not None else dw), (None if (acc_b is not None or db is None) else db)


_HEAD_RIDE = os.environ.get("LIGHTGRAD_HEAD_RIDE", "1") != "0"      # experiments: 0 = head_bwd with its slab workgroups, always
_HEAD_GRAD_AHEAD = os.environ.get("LIGHTGRAD_HEAD_GRAD_AHEAD", "1") != "0"      # experiments: 0 = g_pre by the backward pass, always


def _head_grad_ahead_of(g2, src, weight):
    """(dx, relu.backward's result) that the forward launch wrote ahead (head_mse_forward), if they are what this backward pass needs:
    g2 IS the err of that launch (mse.backward hands it on untouched when its seed is the constant 1), the layer is the same, and
    nobody has written into err, the pre-activation or the weight since (their storage still carries the launch's token)"""
    ahead = g2._head_grad_ahead
    if ahead is None:
        return None
    pre, w, dx, gpre, token = ahead
    if pre is not src or w is not weight:
        return None
    for buf in (src._data, weight._data, g2._data):
        if buf is None or buf.derived is None or buf.derived.get("head_grad") is not token:
            return None
    return dx, gpre


def _head_weight_gradient_can_ride(x, src, relu, weight, bias, acc_w, acc_b, want_db) -> bool:
    """may dW (+ db) of this skinny output layer wait for the backward of the layer in front of it and share ITS launch?  Yes when
    that layer is a `linear` node whose backward launches dW and dx together (linear._backward, `paired`), when the gradients
    are ADDED INTO buffers that exist (parameters after zero_grad: nothing to hand back to the tape, nobody reads them before the
    pass ends) and nobody waits for the moment they are enqueued (DataParallel's hooks) - the MNIST MLP of the headline, and any
    `Linear -> relu -> Linear(<= 16) -> loss` tail."""
    if not (_HEAD_RIDE and relu) or HeldPair.held or GradGroup.active or GradGroup.issuing:
        return False
    if acc_w is None or (bias is not None and not want_db) or (want_db and acc_b is None) or not x.requires_grad:
        return False
    if weight._grad_written_hook is not None or (bias is not None and bias._grad_written_hook is not None):
        return False
    node = src._ctx
    if node is None or node.__class__ is not linear or len(node._parents) > 3 and node._parents[3] is not None:
        return False
    x1, w1 = node._parents[0], node._parents[1]
    b1 = node._parents[2] if len(node._parents) > 2 else None
    if not (isinstance(x1, HipTensor) and x1.requires_grad and w1.requires_grad and len(x1._shape) == 2):
        return False
    if x1._data is None and _lazy_relu_input(x1) is None:       # (a lazy relu goes through _linear_backward_through_lazy_relu: paired as well)
        return False
    if w1._grad_written_hook is not None or (b1 is not None and b1._grad_written_hook is not None):
        return False
    return _rowsum_column_is_cheap(weight._shape[0], weight._shape[1])


def _head_backward_riding(x, src, relu, weight, bias, g2, acc_w, acc_b, want_db):
    """_head_backward with the weight gradient leaving later: dx / g_pre from head_bwd's tile workgroups now (the next tape node
    needs them), dW (+ db) = g^T @ relu(src) prepared as an MFMA product and held (HeldPair) for the launch of the hidden layer's
    two products; the loss of the forward pass is finished by a spare workgroup of that launch."""
    rows, hidden = src._shape
    outs = weight._shape[0]
    lib = _l.lib()
    _l.check(lib.lg_gemm_pair_begin())
    try:
        _gemm_fused(_swap_last(g2), src, relu_b=True, accumulate_into=acc_w, overwrite=weight._consume_zero_pending(),
                    want_rowsum=want_db, rowsum_into=acc_b, rowsum_overwrite=want_db and bias._consume_zero_pending())
        loss = g2._unfinished_loss() if g2._unfinished_loss is not None else None
        row_loss = loss_out = None
        if loss is not None and loss._data is None and loss._lazy_source is not None and loss._lazy_source[0] == "mse_rows" \
                and loss._lazy_source[1]._shape == (rows,) and loss._lazy_source[2] == rows * outs:
            row_loss, loss_out = loss._lazy_source[1], HipTensor.empty((), requires_grad=False)
            _l.check(lib.lg_gemm_pair_mse_loss(row_loss.ptr, rows, rows * outs, loss_out.ptr))
            loss._data, loss._offset, loss._byte_offset, loss._lazy_source = loss_out._data, loss_out._offset, loss_out._byte_offset, None
            g2._unfinished_loss = None
    except BaseException:
        _l.lib().lg_gemm_pair_end()
        raise
    HeldPair.hold(reads=(g2, src, row_loss), writes=(acc_w, acc_b, loss_out))
    ahead = _head_grad_ahead_of(g2, src, weight)
    if ahead is not None:
        dx, gpre = ahead           # nothing to launch: the forward pass wrote both
    else:
        dx = HipTensor.empty((rows, hidden))
        gpre = HipTensor.empty((rows, hidden))
        _l.check(lib.lg_head_bwd_f32(src.ptr, hidden, 1, g2.ptr, weight.ptr, dx.ptr, gpre.ptr, None, 0, None, 0,
                                     rows, hidden, outs, None, None))
    dx._relu_bwd_done = (src, gpre)
    if len(x._shape) != 2:
        dx = dx.reshape(*x._shape)
    return (dx, None) if bias is None else (dx, None, None)


gelu = HipTensor.register_op("gelu", _unary_op("gelu", _l.EW_GELU, _l.EW_GELU_BWD, False,
                                               "tanh-approximated gelu of examples/bert.py:12 as one kernel (fwd) / one kernel (bwd)"))


def _rows_view(t):
    """dense (rows, cols) pointer view of a tensor whose last axis is the row"""
    t = t.contiguous()
    cols = t._shape[-1] if len(t._shape) else 1
    return t, (t.numel() // cols if cols else 0), cols


@HipTensor.register_op(overwrite=True)
class softmax(Function):
    """ row-wise fused softmax (composite: autograd/ops.py:62-66: exp(t - max) * (sum ** -1)); any axis is moved last """
    def forward(ctx, t, axis=-1, scale=1.0):
        _require_f32(t)
        nd = len(t._shape)
        axis = axis % nd
        perm = None
        if axis != nd - 1:
            perm = tuple(i for i in range(nd) if i != axis) + (axis,)
            t = t.transpose(*perm)
        x, rows, cols = _rows_view(t)
        y = HipTensor.empty(x._shape)
        _l.check(_l.lib().lg_softmax_scaled_f32(x.ptr, y.ptr, rows, cols, float(scale)))
        ctx.save_for_backward(_saved_output(y), perm, float(scale))
        if perm is not None:
            inv = [0] * nd
            for i, j in enumerate(perm):
                inv[j] = i
            y = HipTensor(y.data, tuple(y._shape[i] for i in inv), tuple(y._strides[i] for i in inv), y._offset, y._dtype)
        return y

    def backward(ctx, out_grad):
        y, perm, scale = ctx.get_saved_tensors()
        g = out_grad.transpose(*perm) if perm is not None else out_grad
        g, rows, cols = _rows_view(g)
        dx = HipTensor.empty(y._shape)
        _l.check(_l.lib().lg_softmax_scaled_bwd_f32(y.ptr, g.ptr, dx.ptr, rows, cols, scale))
        if perm is not None:
            inv = [0] * len(perm)
            for i, j in enumerate(perm):
                inv[j] = i
            dx = dx.transpose(*inv)
        return dx


def _scaled_softmax(t, scale, axis=-1):
    """softmax(t * scale) in one kernel forward, one backward (attention scores: reference examples/bert.py:81-86 spells it
    `(q @ k / sqrt(d)).softmax(-1)`, three kernels each way); same bits as `(t * scale).softmax(axis)` on this backend"""
    return softmax(t, axis=axis, scale=scale)


HipTensor.scaled_softmax = _scaled_softmax


def _token_rows(t):
    """(tensor, row pitch, batch pitch) of a (batch, positions, width) tensor whose rows the attention kernels can address as
    they lie: width contiguous, pitches multiples of 4, 16-byte aligned; anything else is copied once"""
    st, sh = t._strides, t._shape
    if st[2] != 1 or st[1] % 4 or st[0] % 4 or st[1] < sh[2] or t._byte_offset % 16:
        t = t.contiguous()
        st = t._strides
    return t, st[1], st[0]


def attention_supported(q, heads):
    """does `q.attention(k, v, heads, scale)` exist for this shape? (b, s, heads * d) with d = 32 or 64 and s = 32 .. 128 in 32s"""
    return len(q._shape) == 3 and q._dtype == _F32 and q._shape[2] % heads == 0 and \
        bool(_l.lib().lg_attention_supported(q._shape[1], q._shape[2] // heads))


@HipTensor.register_op()
class attention(Function):
    """ softmax((q k^T) * scale) v per head, forward and backward in one launch each (csrc/attention.hip); q, k, v are the
    (batch, positions, heads * d) outputs of the three projections as they stand - the head split of examples/bert.py:78-80
    happens in the kernels' addressing.  The probabilities (batch, heads, s, s) the reference model returns next to the context
    (bert.py:88) are on the result as `.attention_probs`, outside the tape (the composite form differentiates through them) """
    def forward(ctx, q, k, v, heads=1, scale=1.0):
        _require_f32(q, k, v)
        assert q._shape == k._shape == v._shape and attention_supported(q, heads), \
            "attention: unsupported shapes %s / %s / %s with %d heads" % (q._shape, k._shape, v._shape, heads)
        b, s, width = q._shape
        d = width // heads
        (q, ldq, sbq), (k, ldk, sbk), (v, ldv, sbv) = _token_rows(q), _token_rows(k), _token_rows(v)
        out = HipTensor.empty((b, s, width))
        probs = HipTensor.empty((b, heads, s, s), requires_grad=False)
        _l.check(_l.lib().lg_attention_fwd_f32(q.ptr, ldq, sbq, k.ptr, ldk, sbk, v.ptr, ldv, sbv, out.ptr, width, s * width,
                                               probs.ptr, b, heads, s, d, float(scale)))
        ctx.save_for_backward(q, k, v, probs, heads, float(scale))
        out.attention_probs = probs
        return out

    def backward(ctx, out_grad):
        q, k, v, probs, heads, scale = ctx.get_saved_tensors()
        b, s, width = q._shape
        (q, ldq, sbq), (k, ldk, sbk), (v, ldv, sbv), (g, ldg, sbg) = _token_rows(q), _token_rows(k), _token_rows(v), _token_rows(out_grad)
        dq, dk, dv = HipTensor.empty((b, s, width)), HipTensor.empty((b, s, width)), HipTensor.empty((b, s, width))
        _l.check(_l.lib().lg_attention_bwd_f32(q.ptr, ldq, sbq, k.ptr, ldk, sbk, v.ptr, ldv, sbv, g.ptr, ldg, sbg, probs.ptr,
                                               dq.ptr, width, s * width, dk.ptr, width, s * width, dv.ptr, width, s * width,
                                               b, heads, s, width // heads, scale))
        return dq, dk, dv


HipTensor.attention_supported = attention_supported


def self_attention_supported(x, wq, heads):
    """does `x.self_attention(wq, bq, wk, bk, wv, bv, heads, scale)` exist for these shapes?  (b, s, hidden) input, three
    (width, hidden) weights with width = heads * d, d = 32 or 64, width a multiple of 64, s = 32 .. 128 in 32s"""
    return (len(x._shape) == 3 and x._dtype == _F32 and len(wq._shape) == 2 and wq._shape[1] == x._shape[2] and wq._shape[0] % heads == 0
            and wq._shape[0] % 64 == 0 and x._shape[2] % 4 == 0 and x.numel() > 0
            and bool(_l.lib().lg_attention_supported(x._shape[1], wq._shape[0] // heads)))


def _ptr3(a, b, c):
    return (ctypes.c_void_p * 3)(a, b, c)


@HipTensor.register_op()
class self_attention(Function):
    """ the query / key / value projections and the attention over them as ONE tape node (reference examples/bert.py:78-88:
    three nn.Linear, scores, scaling, softmax, context): the three projections are one launch (lg_gemm_multi3_f32 - the weights
    stay the separately allocated parameters they are) into one (b, s, 3 * width) buffer that the attention kernel reads in
    place; backward: the attention kernel writes dq | dk | dv into one buffer of that shape, the input gradient is ONE product
    whose K runs through the three weights (lg_gemm_kseg3_f32, added to a gradient the input already holds in its epilogue),
    the weight / bias gradients take the routes of `linear`.  `.attention_probs` as for `attention`. """
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, heads=1, scale=1.0):
        _require_f32(x, wq, bq, wk, bk, wv, bv)
        assert self_attention_supported(x, wq, heads) and wq._shape == wk._shape == wv._shape and bq._shape == bk._shape == bv._shape == (wq._shape[0],), \
            "self_attention: unsupported shapes %s with weights %s / %s / %s and %d heads" % (x._shape, wq._shape, wk._shape, wv._shape, heads)
        b, s, hidden = x._shape
        width = wq._shape[0]
        x = x.contiguous()
        ws, bs = [w.contiguous() for w in (wq, wk, wv)], [t.contiguous() for t in (bq, bk, bv)]
        qkv = HipTensor.empty((b, s, 3 * width), requires_grad=False)
        base = qkv.ptr
        _l.check(_l.lib().lg_gemm_multi3_f32(0, 1, b * s, width, hidden, x.ptr, hidden, _ptr3(*(w.ptr for w in ws)), hidden,
                                             _ptr3(base, base + 4 * width, base + 8 * width), 3 * width, _ptr3(*(t.ptr for t in bs))))
        out = HipTensor.empty((b, s, width))
        probs = HipTensor.empty((b, heads, s, s), requires_grad=False)
        ld, sb = 3 * width, s * 3 * width
        _l.check(_l.lib().lg_attention_fwd_f32(base, ld, sb, base + 4 * width, ld, sb, base + 8 * width, ld, sb, out.ptr, width, s * width,
                                               probs.ptr, b, heads, s, width // heads, float(scale)))
        ctx.save_for_backward(x, qkv, probs, heads, float(scale))
        out.attention_probs = probs
        return out

    def backward(ctx, out_grad):
        x, qkv, probs, heads, scale = ctx.get_saved_tensors()
        x_in = ctx._parents[0]
        params = ctx._parents[1:7]
        b, s, hidden = x._shape
        width = qkv._shape[2] // 3
        g, ldg, sbg = _token_rows(out_grad)
        dqkv = HipTensor.empty((b, s, 3 * width), requires_grad=False)
        base, dbase = qkv.ptr, dqkv.ptr
        ld, sb = 3 * width, s * 3 * width
        _l.check(_l.lib().lg_attention_bwd_f32(base, ld, sb, base + 4 * width, ld, sb, base + 8 * width, ld, sb, g.ptr, ldg, sbg, probs.ptr,
                                               dbase, ld, sb, dbase + 4 * width, ld, sb, dbase + 8 * width, ld, sb,
                                               b, heads, s, width // heads, scale))
        x2 = x.reshape(-1, hidden)
        grads = []
        for i in range(3):
            weight, bias = params[2 * i], params[2 * i + 1]
            gi = HipTensor(dqkv.data, (b * s, width), (3 * width, 1), dqkv._offset + i * width, _F32, requires_grad=False)
            want_db = bias.requires_grad
            acc_w = weight._grad_accumulator() if weight.requires_grad else None
            acc_w = acc_w if (acc_w is not None and acc_w.is_contiguous()) else None
            acc_b = bias._grad_accumulator() if want_db else None
            acc_b = acc_b if (acc_b is not None and acc_b.is_contiguous()) else None
            if acc_w is not None and (not want_db or acc_b is not None) and GradGroup.usable_for(weight, bias):
                with GradGroup.issue(reads=(gi, x2), writes=(acc_w, acc_b)):
                    grads += list(linear._weight_products(x2, weight, bias, want_db, gi, acc_w, acc_b))
            else:
                grads += list(linear._weight_products(x2, weight, bias, want_db, gi, acc_w, acc_b))
        dx = None
        if x_in.requires_grad:
            ws = [params[0].contiguous(), params[2].contiguous(), params[4].contiguous()]
            wptrs = _ptr3(*(w.ptr for w in ws))
            have = x_in._grad if (x_in._ctx is not None and x_in._view_of_leaf is None) else None
            if (have is not None and have.__class__ is HipTensor and have._shape == x_in._shape and have._dtype == _F32 and have.is_contiguous()):
                # the input already holds a contribution (the residual branch): added in this product's epilogue
                if x_in._grad_shared:
                    new = HipTensor.empty(x_in._shape, requires_grad=False)
                    _l.check(_l.lib().lg_gemm_kseg3_f32(0, 0, b * s, hidden, width, dbase, 3 * width, wptrs, hidden, new.ptr, hidden, 0,
                                                        have.ptr, hidden))
                    x_in._grad, x_in._grad_shared = new, False
                else:
                    flush_lazy_readers(have)
                    _l.check(_l.lib().lg_gemm_kseg3_f32(0, 0, b * s, hidden, width, dbase, 3 * width, wptrs, hidden, have.ptr, hidden, 1, None, 0))
            else:
                dx = HipTensor.empty(x_in._shape, requires_grad=False)
                _l.check(_l.lib().lg_gemm_kseg3_f32(0, 0, b * s, hidden, width, dbase, 3 * width, wptrs, hidden, dx.ptr, hidden, 0, None, 0))
        return (dx,) + tuple(grads)


HipTensor.self_attention_supported = self_attention_supported


@HipTensor.register_op()
class layer_norm(Function):
    """ nn.LayerNorm over the last axis in one kernel (composite: nn.py:109-124); backward dx fused,
    dw = sum_rows(g * xhat), db = sum_rows(g) """
    def forward(ctx, x, weight, bias, eps=1e-5):
        _require_f32(x, weight, bias)
        assert len(weight._shape) == 1 and weight._shape == bias._shape == x._shape[-1:], \
            "layer_norm fuses a 1-D normalised shape; got %s for input %s" % (weight._shape, x._shape)
        xc, rows, cols = _rows_view(x)
        w, b = weight.contiguous(), bias.contiguous()
        y, xhat, rstd = HipTensor.empty(xc._shape), HipTensor.empty(xc._shape), HipTensor.empty((rows,))
        _l.check(_l.lib().lg_layernorm_f32(xc.ptr, w.ptr, b.ptr, y.ptr, xhat.ptr, rstd.ptr, rows, cols, float(eps)))
        ctx.save_for_backward(w, xhat, rstd, rows, cols)
        return y

    def backward(ctx, out_grad):
        w, xhat, rstd, rows, cols = ctx.get_saved_tensors()
        g = out_grad.contiguous()
        dx = HipTensor.empty(xhat._shape)
        _l.check(_l.lib().lg_layernorm_bwd_f32(g.ptr, w.ptr, xhat.ptr, rstd.ptr, dx.ptr, rows, cols))
        # dw = sum_rows(g * xhat), db = sum_rows(g) from one launch, added straight into the parameters' gradient buffers
        # when they have one (leaf parameters after zero_grad), like linear.backward does
        weight, bias = ctx._parents[1], ctx._parents[2]
        acc_w = weight._grad_accumulator() if weight.requires_grad else None
        acc_b = bias._grad_accumulator() if bias.requires_grad else None
        acc_w = acc_w if (acc_w is not None and acc_w.is_contiguous()) else None
        acc_b = acc_b if (acc_b is not None and acc_b.is_contiguous()) else None
        dw = acc_w if acc_w is not None else HipTensor.empty((cols,))
        db = acc_b if acc_b is not None else HipTensor.empty((cols,))

        def param_grads():
            for t in (acc_w, acc_b):
                if t is not None:
                    flush_lazy_readers(t)
            _l.check(_l.lib().lg_layernorm_param_grads_f32(
                g.ptr, xhat.ptr, dw.ptr, db.ptr, rows, cols,
                1 if (acc_w is not None and not weight._consume_zero_pending()) else 0,
                1 if (acc_b is not None and not bias._consume_zero_pending()) else 0))
        if acc_w is not None and acc_b is not None and GradGroup.usable_for(weight, bias):
            with GradGroup.issue(reads=(g, xhat), writes=(acc_w, acc_b)):          # queued, like a Linear's dW
                param_grads()
        else:
            param_grads()
        if acc_w is not None:
            weight._notify_grad_written()
        if acc_b is not None:
            bias._notify_grad_written()
        return dx, (None if acc_w is not None else dw), (None if acc_b is not None else db)


def _gather_rows(table, ids):
    """table[ids]: ids is an int32/int64 HipTensor indexing the first axis (embedding lookup)"""
    _require_f32(table)
    assert ids._dtype in (np.dtype(np.int32), np.dtype(np.int64)), "index tensor must be int32 or int64, got %s" % ids._dtype
    table, ids = table.contiguous(), ids.contiguous()
    row_len = 1
    for s in table._shape[1:]:
        row_len *= s
    out = HipTensor.empty(ids._shape + table._shape[1:])
    _l.check(_l.lib().lg_gather_rows_f32(table.ptr, ids.ptr, ids._dtype.itemsize, out.ptr, ids.numel(), row_len, table._shape[0]))
    return out


def _scatter_add_rows(shape, ids, out_grad, into=None):
    if into is not None:
        flush_lazy_readers(into)
    grad = into if into is not None else HipTensor.zeros(shape, requires_grad=False)
    ids, g = ids.contiguous(), out_grad.contiguous()
    row_len = 1
    for s in shape[1:]:
        row_len *= s
    _l.check(_l.lib().lg_scatter_add_rows_f32(g.ptr, ids.ptr, ids._dtype.itemsize, grad.ptr, ids.numel(), row_len, shape[0]))
    return grad


def mse_forward(y, y_hat):
    """fused loss.mse forward: returns (loss of shape (), err = y - y_hat) from one pass (lg_mse_f32)"""
    _require_f32(y, y_hat)
    assert y._shape == y_hat._shape, "mse: shapes %s and %s differ" % (y._shape, y_hat._shape)
    y, y_hat = y.contiguous(), y_hat.contiguous()
    err, loss = HipTensor.empty(y._shape), HipTensor.empty(())
    _l.check(_l.lib().lg_mse_f32(y.ptr, y_hat.ptr, err.ptr, loss.ptr, y.numel()))
    return loss, err


def cross_entropy_forward(y, labels):
    """fused loss.cross_entropy forward for logits (N, C) and integer labels (N,): returns (mean nll of shape (),
    dlogits = (softmax(y) - onehot) / N) from one row-wise pass (lg_cross_entropy_f32) + the mean over rows"""
    _require_f32(y)
    assert len(y._shape) == 2 and labels._shape == (y._shape[0],), \
        "cross_entropy: logits %s and labels %s do not match" % (y._shape, labels._shape)
    assert labels._dtype in (np.int16, np.int32, np.int64), "cross_entropy: labels must be int16/int32/int64"
    y, labels = y.contiguous(), labels.contiguous()
    n, c = y._shape
    dlogits, nll = HipTensor.empty(y._shape), HipTensor.empty((n,))
    if n == 0:
        _l.check(_l.lib().lg_cross_entropy_f32(y.ptr, labels.ptr, labels._dtype.itemsize, dlogits.ptr, nll.ptr, n, c))
        total = _reduce(_l.RED_SUM, nll, (0,), False)
        return _ew(_l.EW_MUL, (), [total, None], scalar=float("nan")), dlogits          # mean of nothing
    loss = HipTensor.empty(())
    _l.check(_l.lib().lg_cross_entropy_mean_f32(y.ptr, labels.ptr, labels._dtype.itemsize, dlogits.ptr, nll.ptr, loss.ptr, n, c))
    return loss, dlogits


def adam_step_(p, g, m, v, lr, b1, b2, eps, inv_bias1, inv_bias2, gscale=1.0, belief=False):
    """in-place fused Adam/AdaBelief update of dense fp32 tensors (lg_adam_step_f32)"""
    _require_f32(p, g, m, v)
    for t in (p, g, m, v):
        assert t.is_contiguous() and t._shape == p._shape, "adam_step_ needs dense tensors of one shape"
    flush_lazy_readers(p)
    _l.check(_l.lib().lg_adam_step_f32(p.ptr, g.ptr, m.ptr, v.ptr, p.numel(), lr, b1, b2, eps, inv_bias1, inv_bias2, gscale,
                                       1 if belief else 0))
