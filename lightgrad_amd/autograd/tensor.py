"""Backend-independent tensor base: data handle, gradient, tape context,
op registry and backend registry.

Behavioural restatement of the reference's `lightgrad/autograd/tensor.py`:
  * `_TensorType` registers a converter `AbstractTensor.<backend>()` for every
    tensor class defined in a module `<pkg>.autograd.<backend>.tensor`
    (tensor.py:5-15, :154-161) - this is how `.cpu()` and `.hip()` appear
  * `backward` seeds `ones(shape)` and only starts from item tensors unless
    `allow_fill` (tensor.py:99-109)
  * `add_grad` copies on first touch, `+=` afterwards (tensor.py:111-118)
  * `zero_grad` allocates zeros or `fill(0)`s the existing gradient and can
    walk the graph (tensor.py:120-131)
  * `register_op` installs a dispatching method; refuses non-Functions
    (TypeError) and silent re-registration (RuntimeError) (tensor.py:136-152)
"""
import numpy as np
from .grads import Gradients


class _TensorType(type):

    def __new__(mcs, name, bases, attrs):
        T = type.__new__(mcs, name, bases, attrs)
        module = attrs.get('__module__')
        # classes of this module (the abstract base) and classes minted at run time are skipped
        if (module is not None) and (module != __name__):
            parts = module.split('.')
            if len(parts) >= 2:
                AbstractTensor.register_backend(parts[-2], T)
        return T


class AbstractTensor(metaclass=_TensorType):

    # A backend may set this to let NON-LEAF tensors adopt the first gradient they receive instead of
    # copying it (the copy of tensor.py:116 is kept for leaves).  The adopted tensor is treated as
    # read-only: a second contribution makes a fresh sum, zero_grad replaces instead of filling.
    # Values are identical to the copying form; it only removes one device copy per tape node.
    _adopt_first_grad = False

    # Set by an optimizer's zero_grad in place of the fill: "this gradient buffer counts as zeros".  The first kernel
    # that would accumulate into it overwrites it instead (`_consume_zero_pending`); anything else that looks at the
    # gradient first gets the zeros written for real (`grad`, `add_grad`, `_materialize_zero_grad`).  Same values as
    # fill(0) followed by +=, one pass over the buffer less.
    _grad_zero_pending = False

    # Optional callable(tensor) invoked every time a gradient contribution for this LEAF has been written or enqueued
    # (`add_grad`, or a backward kernel that accumulated straight into `_grad_accumulator()`).  dist.DataParallel sets
    # it on parameters to start the gradient exchange as soon as the last one is there, while backward still runs.
    _grad_written_hook = None

    def __init__(self, data, requires_grad: bool = True) -> None:
        self._data = data
        self._grad = None
        self._grad_shared = False
        self._requires_grad = requires_grad
        self._ctx = None

    def _set_ctx(self, ctx) -> "AbstractTensor":
        assert (ctx is None) or isinstance(ctx, Function)
        self._ctx = ctx
        return self

    def _set_data(self, data) -> "AbstractTensor":
        self._data = data
        return self

    def detach(self) -> "AbstractTensor":
        # like the reference this cuts the tape in place (no copy), tensor.py:35-38
        self._ctx = None
        return self

    @property
    def ctx(self):
        return self._ctx

    @property
    def data(self):
        return self._data

    @property
    def grad(self) -> "AbstractTensor":
        if self._grad_zero_pending:
            self._materialize_zero_grad()
        return self._grad

    def _materialize_zero_grad(self) -> None:
        if self._grad_zero_pending:
            self._grad_zero_pending = False
            self._grad.fill(0)

    def _consume_zero_pending(self) -> bool:
        """True once after a lazy zero_grad: the caller then WRITES its gradient into `_grad_accumulator()`"""
        pending, self._grad_zero_pending = self._grad_zero_pending, False
        return pending

    @property
    def requires_grad(self) -> bool:
        return self._requires_grad

    @property
    def dtype(self):
        raise NotImplementedError()

    @property
    def shape(self) -> tuple:
        raise NotImplementedError()

    def item(self):
        return self.numpy().item()

    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return int(n)

    """ Initializers """

    @staticmethod
    def empty(shape, requires_grad: bool = True) -> "AbstractTensor":
        raise NotImplementedError()

    @staticmethod
    def zeros(shape, requires_grad: bool = True) -> "AbstractTensor":
        raise NotImplementedError()

    @staticmethod
    def ones(shape, requires_grad: bool = True) -> "AbstractTensor":
        raise NotImplementedError()

    @staticmethod
    def uniform(low, high, shape, requires_grad: bool = True) -> "AbstractTensor":
        raise NotImplementedError()

    @staticmethod
    def from_numpy(a: np.ndarray, requires_grad: bool = True) -> "AbstractTensor":
        raise NotImplementedError()

    @classmethod
    def xavier(cls, shape, requires_grad: bool = True) -> "AbstractTensor":
        """U(-1,1) / sqrt(numel), scaled in place (reference tensor.py:85-89)."""
        t = cls.uniform(-1, 1, shape=shape, requires_grad=requires_grad)
        t /= np.sqrt(t.numel())
        return t.detach()

    def copy(self, requires_grad: bool = True) -> "AbstractTensor":
        raise NotImplementedError()

    def numpy(self) -> np.ndarray:
        raise NotImplementedError()

    """ Gradients """

    def backward(self, allow_fill: bool = False) -> None:
        if self._ctx is None:
            return
        if self.shape == (1,) or len(self.shape) == 0 or allow_fill:
            self._grad, self._grad_shared = self._seed_gradient()
        else:
            raise RuntimeError("Can only backpropagate from item tensors!")
        Gradients.backward(self._ctx, self._grad)

    def _seed_gradient(self):
        """(gradient the backward pass starts from, is-it-shared): ones(shape) (reference tensor.py:105).  A backend
        may hand out a cached read-only constant for item tensors by returning shared=True."""
        return self.__class__.ones(self.shape, requires_grad=False), False

    def add_grad(self, grad: "AbstractTensor") -> None:
        if not self._requires_grad:
            return
        Gradients.disable()
        try:
            if self._grad is None:
                if self._adopt_first_grad and self._ctx is not None:
                    self._grad, self._grad_shared = grad, True
                else:
                    self._grad = grad.copy(requires_grad=False)
            elif self._grad_shared:
                self._grad, self._grad_shared = self._grad + grad, False
            else:
                self._materialize_zero_grad()
                self._grad += grad
        finally:
            Gradients.enable()
        if self._grad_written_hook is not None:
            self._grad_written_hook(self)

    def _notify_grad_written(self) -> None:
        """for backward ops that accumulated into `_grad_accumulator()` themselves and report None (func.py extension)"""
        if self._grad_written_hook is not None:
            self._grad_written_hook(self)

    def _grad_accumulator(self):
        """The gradient buffer a backward op may add into directly, or None.  Only LEAF tensors that already own a
        gradient qualify (parameters after zero_grad): for them `add_grad(g)` is exactly `self.grad += g`
        (tensor.py:118), which a kernel with an accumulate flag does without materialising g."""
        if self._requires_grad and self._ctx is None and self._grad is not None and not self._grad_shared:
            return self._grad
        return None

    def zero_grad(self, traverse_graph: bool = False) -> None:
        if not traverse_graph:
            self._zero_own_grad()
            return
        # walk every ancestor ONCE (the reference recurses, tensor.py:127-131, which revisits shared sub-graphs
        # and is exponential on residual networks; zeroing is idempotent, so a visited set changes nothing else)
        seen, stack = {id(self)}, [self]
        while stack:
            t = stack.pop()
            t._zero_own_grad()
            if t._ctx is not None:
                for p in t._ctx.parent_tensors:
                    assert p is not t
                    if id(p) not in seen:
                        seen.add(id(p))
                        stack.append(p)

    def _zero_own_grad(self) -> None:
        self._grad_zero_pending = False
        if self._requires_grad:
            if self._grad is None or self._grad_shared:
                self._grad, self._grad_shared = self.__class__.zeros(self.shape, requires_grad=False), False
            elif self._ctx is None:
                # LAZY for a LEAF that already owns its gradient buffer: the first backward kernel that reaches it overwrites
                # (GEMM with beta = 0: every user of `_grad_accumulator()` asks `_consume_zero_pending()`), `add_grad` and every
                # reader (`.grad`) fill first (`_materialize_zero_grad`).  Saves one pass over the buffer here and the read of the
                # zeros in the accumulating epilogue: 4096^2 matmul forward+backward with its two 64 MiB gradients 139.7 -> 143.7
                # TFLOP/s together with the register work on the large tile.  NOT for intermediates: backward ops add into an
                # intermediate's `_grad` directly (the residual branch of `linear._input_product`) - a pending fill would wipe it
                # at the next read (found by gradcheck's repeated backward passes over one tape).
                self._grad_zero_pending = True
            else:
                self._grad.fill(0)

    """ Registration of operations and backends """

    @classmethod
    def register_op(cls, name: str = None, op: type = None, overwrite: bool = False):
        if op is None:
            # decorator form
            return lambda fn_cls: cls.register_op(name if name is not None else fn_cls.__name__, fn_cls, overwrite=overwrite)
        if not (isinstance(op, type) and issubclass(op, Function)):
            raise TypeError("Operators must inherit from Function! (%s)" % getattr(op, '__name__', op))
        if not overwrite and hasattr(cls, name):
            raise RuntimeError("Function %s already registered to %s!" % (name, cls.__name__))

        # a plain function attribute binds as a method; the Function class itself would not
        def dispatch(self, *args, **kwargs):
            return op(self, *args, **kwargs)
        dispatch.__name__ = name
        dispatch.__doc__ = op.__doc__
        setattr(cls, name, dispatch)
        return op

    @staticmethod
    def register_backend(name: str, tensor_cls: type):
        if not issubclass(tensor_cls, AbstractTensor):
            raise TypeError("Backend tensors must inherit from Tensor! (%s)" % tensor_cls.__name__)

        def convert(t, *args, **kwargs):
            return tensor_cls.from_numpy(t.numpy(), *args, **kwargs)
        convert.__name__ = name
        setattr(AbstractTensor, name, convert)


# bottom imports: func needs AbstractTensor, ops registers the composite operators on it
from .func import Function  # noqa: E402
from . import ops  # noqa: E402,F401
