"""Tape nodes: `Function` (first-class op with hand-written backward) and
`WrapperFunction` (composite op differentiated through an internal sub-tape).

Behavioural restatement of the reference's `lightgrad/autograd/func.py`:
  * calling a Function *class* runs `forward` under no_grad and returns the
    output tensor with the node attached as `ctx`          (func.py:11-29)
  * all positional arguments are recorded as parents        (func.py:15-16, :32-33)
  * tensors of one call must share a backend class          (func.py:18-20)
  * `_backpropagate` un-broadcasts gradients with `sum(axis, keepdims=True)` +
    `reshape` before `add_grad`                             (func.py:40-59)
  * a Function without `backward` raises
    RuntimeError("Cannot Backward through X!")              (func.py:68-69)
  * WrapperFunction runs `forward` with gradients enabled, stashes the inner
    root ctx and replays it in `_backpropagate`             (func.py:71-106)
"""
from .grads import Gradients
from .utils.profiler import Tracker, Profiler


class _FunctionType(type):

    def _apply(cls, f, args, kwargs):
        # first-class ops never record their internals
        Gradients.disable()
        try:
            return f.forward(*args, **kwargs)
        finally:
            Gradients.enable()

    def __call__(cls, *args, **kwargs):
        f = object.__new__(cls)
        f.__init__(*args)
        # one backend per call
        tensor_type = None
        for t in args:
            if isinstance(t, AbstractTensor):
                if tensor_type is None:
                    tensor_type = t.__class__
                elif t.__class__ is not tensor_type:
                    assert isinstance(t, tensor_type), \
                        "All Tensors must be of the same type! %s" % str(
                            tuple(x.__class__.__name__ for x in args if isinstance(x, AbstractTensor)))
        if kwargs:
            for v in kwargs.values():
                if isinstance(v, AbstractTensor):
                    # tensors passed by keyword are not parents and therefore must not need gradients
                    assert not v.requires_grad, "Tensors that require gradients must be passed positionally!"
                    assert tensor_type is None or isinstance(v, tensor_type), "All Tensors must be of the same type!"
        if Profiler._active_profilers:
            with Tracker(cls.__name__):
                out = cls._apply(f, args, kwargs)
        else:
            out = cls._apply(f, args, kwargs)
        assert isinstance(out, AbstractTensor)
        if Gradients._disable_depth == 0:
            if f.parent_tensors:
                out._set_ctx(f)
            else:
                # extension of the reference (func.py:25-28 attaches the node regardless): nothing upstream wants a
                # gradient (inputs created with requires_grad=False), so the result is a constant of the tape as well -
                # no node, and backward ops further down skip the gradient for it (e.g. dx of the first Linear)
                out._requires_grad = False
        return out


class Function(object, metaclass=_FunctionType):

    def __init__(self, *parents):
        self._parents = parents
        self._saved = ()
        self._wanting = None

    @property
    def parent_tensors(self):
        """Positional parents that are tensors requiring gradients (settled when the node is made: asked for several times per
        node by the call protocol and by the backward walk)."""
        wanting = self._wanting
        if wanting is None:
            wanting = self._wanting = [t for t in self._parents if isinstance(t, AbstractTensor) and t._requires_grad]
        return wanting

    def _backpropagate(self, out_grad):
        if Profiler._active_profilers:
            with Tracker(self.__class__.__name__, backward=True):
                self._backpropagate_impl(out_grad)
        else:
            self._backpropagate_impl(out_grad)

    def _backpropagate_impl(self, out_grad):
        in_grads = self.backward(out_grad)
        if not isinstance(in_grads, tuple):
            in_grads = (in_grads,)
        for t, g in zip(self._parents, in_grads):
            if not (isinstance(t, AbstractTensor) and t.requires_grad):
                continue
            if g is None:
                # extension of the reference protocol (func.py:47 asserts non-None): a backward may add a gradient
                # straight into `t._grad_accumulator()` (e.g. a GEMM with beta = 1) and report None for that parent
                continue
            if g.shape != t.shape:
                g = _unbroadcast(g, t.shape)
            assert g.shape == t.shape
            t.add_grad(g)

    def save_for_backward(ctx, *objs):
        ctx._saved += tuple(objs)

    def get_saved_tensors(ctx):
        return ctx._saved

    def forward(ctx, t, *args, **kwargs):
        raise NotImplementedError()

    def backward(ctx, out_grad):
        raise RuntimeError("Cannot Backward through %s!" % ctx.__class__.__name__)


def _unbroadcast(g, shape):
    """Reduce a broadcast gradient back to `shape` (reference func.py:50-56)."""
    assert len(g.shape) >= len(shape), "Cannot unbroadcast shapes %s and %s" % (shape, g.shape)
    lead = len(g.shape) - len(shape)
    axes = tuple(range(lead)) + tuple(lead + i for i, (x, y) in enumerate(zip(shape, g.shape[lead:])) if x != y)
    g = g.sum(axis=axes, keepdims=True)
    return g.reshape(*g.shape[lead:])


class _WrapperFunctionType(_FunctionType):

    def _apply(cls, f, args, kwargs):
        # gradients stay enabled: the inner first-class ops build a sub-tape
        out = f.forward(*args, **kwargs)
        # the output's ctx is about to be replaced by the wrapper node
        f._set_internal_ctx(out.ctx)
        return out


class WrapperFunction(Function, metaclass=_WrapperFunctionType):
    """Composite op; its gradient is obtained by replaying the recorded sub-tape."""

    def __init__(self, *parents):
        Function.__init__(self, *parents)
        self._internal_ctx = None

    def _set_internal_ctx(self, ctx):
        self._internal_ctx = ctx

    def _backpropagate_impl(self, out_grad):
        if self._internal_ctx is None:
            # forward ran with gradients disabled or returned an input unchanged
            return
        # fence the sub-tape at the wrapper's parents
        fenced = [(p, p.ctx) for p in self.parent_tensors]
        for p, _ in fenced:
            p._set_ctx(None)
        try:
            # tensors created inside the wrapper are invisible to zero_grad(traverse_graph=True) (it walks
            # the wrapper's parents, not its sub-tape), so their gradients are transient: start clean,
            # otherwise a second backward through the same graph (gradcheck.jacobian) would re-propagate
            # the previous call's gradient.  A single backward is unaffected.
            for node in Gradients._schedule(self._internal_ctx):
                for t in node.parent_tensors:
                    if t.ctx is not None:
                        t._grad, t._grad_shared = None, False
            Gradients.backward(self._internal_ctx, out_grad)
        finally:
            for p, c in fenced:
                p._set_ctx(c)

    def forward(ctx, *args, **kwargs):
        raise NotImplementedError()

    @staticmethod
    def from_function(fn):
        """Decorator: plain python function of tensors -> WrapperFunction subclass."""
        def forward(ctx, *args, **kwargs):
            return fn(*args, **kwargs)
        return type(fn.__name__, (WrapperFunction,), {'forward': forward, '__doc__': fn.__doc__})


from .tensor import AbstractTensor  # noqa: E402  (bottom import: circular dependency)
