"""HipTensor: strided device tensor on an MI355X, the backend this repo adds
next to CpuTensor.

Same shape as the reference's OpenCLTensor (opencl/tensor.py:18-116): a device
buffer plus shape / strides (in elements) / offset, so `transpose` and basic
`__getitem__` views are stride arithmetic and never move data.  Differences by
design: memory comes from liblghip's caching stream-ordered pool, every
operation is asynchronous on one HIP stream (only `numpy()`/`item()` block),
and scalar results have shape `()` like the CPU backend (the reference's OpenCL
tensor coerces `()` to `(1,)`, opencl/tensor.py:35).

There is no CPU fallback anywhere in this class: without liblghip.so and a GPU
every constructor raises `HipError`.
"""
import ctypes
import os
import numpy as np
from ..tensor import AbstractTensor
from . import lib as _l


_strides_cache = {}


def contiguous_strides(shape):
    """row-major strides (in elements) of a dense tensor of this shape; cached - the same few shapes recur every step"""
    st = _strides_cache.get(shape)
    if st is None:
        strides, acc = [], 1
        for s in reversed(shape):
            strides.append(acc)
            acc *= s
        st = tuple(reversed(strides))
        if len(_strides_cache) < 4096:
            _strides_cache[tuple(shape)] = st
    return st


class HipBuffer(object):
    """Owner of one pool allocation; returned to the pool when the last tensor viewing it dies."""
    __slots__ = ("ptr", "nbytes", "lazy_readers", "derived", "frozen", "__weakref__")

    def __init__(self, nbytes):
        self.ptr = None
        self.lazy_readers = None      # weak references to lazy tensors that will still READ this block (see flush_lazy_readers)
        self.derived = None           # tensors computed from this block's CONTENTS and kept for reuse (ops._tiled_ids): dropped by any in-place writer
        self.frozen = False           # HipTensor.freeze(): a constant - in-place writers raise, results derived from it may be recorded in hipGraphs
        p = ctypes.c_void_p()
        _l.check(_l.lib().lg_malloc(ctypes.byref(p), max(int(nbytes), 1)))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    def __del__(self):
        ptr, self.ptr = self.ptr, None
        if ptr is not None and _l._lib is not None:
            _l._lib.lg_free(ptr)


class GradGroup(object):
    """Parameter-gradient kernels of a deep backward pass, queued and launched TOGETHER (lg_gemm_group_* in include/lghip.h).

    In a transformer-sized tape (>= MIN_NODES nodes) the weight gradient of every Linear is a small product with a long K -
    12 us alone, most of it launch, prologue and split-K hand-off on a handful of workgroups - and nothing but the optimizer
    waits for it; the same goes for LayerNorm's weight / bias gradients (7.5 us each).  While the pass runs, those launches
    are queued inside the library (`with GradGroup.issue(reads, writes)`) and go out as one GEMM launch + one LayerNorm launch
    when the pass ends (tiny-BERT: 13 + 6 launches become 1 + 1).  Shallow tapes (the MLP) do not use it:
    there dW and dx share a launch (lg_gemm_pair_*), which is the better deal when the optimizer follows at once.

    Hazards: the queued launches read activations / gradients later - `keep` holds them alive, and an in-place writer into
    their storage or into the gradient buffers (flush_lazy_readers is the hook of every in-place writer) flushes the queue
    first; the library itself flushes before anything that exposes results to the host or to another graph."""
    MIN_NODES = 14          # (tiny-BERT with its fused blocks: about 20 nodes; the MLP of the headline: 6)
    enabled = os.environ.get("LIGHTGRAD_GRAD_GROUP", "1") != "0"
    depth = 0             # nested backward passes (WrapperFunction replays its inner tape with Gradients.backward)
    active = False
    issuing = False
    touched = set()       # id(HipBuffer) of operands and outputs of queued launches
    keep = []

    @staticmethod
    def pass_begins(n_nodes):
        G = GradGroup
        if G.depth == 0:
            G.active = G.enabled and n_nodes >= G.MIN_NODES
        G.depth += 1

    @staticmethod
    def pass_ends():
        G = GradGroup
        G.depth -= 1
        if G.depth == 0 and G.active:
            G.active = False
            G.flush()

    @staticmethod
    def flush():
        G = GradGroup
        del G.keep[:]
        G.touched.clear()
        _l.check(_l.lib().lg_gemm_group_flush())

    @staticmethod
    def usable_for(*params) -> bool:
        """not when someone waits for the moment a gradient's kernel is enqueued (dist.DataParallel starts its exchange from
        that hook)"""
        G = GradGroup
        if not G.active or G.issuing:
            return False
        for p in params:
            if p is not None and p._grad_written_hook is not None:
                return False
        return True

    class issue(object):
        """`with GradGroup.issue(reads, writes): <launches the library may queue>`"""
        __slots__ = ()

        def __init__(self, reads, writes):
            G = GradGroup
            for t in reads:
                if t is not None:
                    G.touched.add(id(t.data))           # (.data: a lazy operand becomes real now, not at the flush)
                    G.keep.append(t)
            for t in writes:
                if t is not None:
                    _make_lazy_readers_real(t)          # snapshots of the old contents
                    G.touched.add(id(t._data))

        def __enter__(self):
            GradGroup.issuing = True
            _l.check(_l.lib().lg_gemm_group_begin())

        def __exit__(self, *exc):
            GradGroup.issuing = False
            _l.check(_l.lib().lg_gemm_group_end())


class HeldPair(object):
    """The weight gradient of a skinny output layer (ops._head_backward), prepared and HELD inside the library
    (lg_gemm_pair_begin ... lg_gemm_pair_hold) until the hidden layer's backward - two tape nodes later - resumes the bracket and
    sends its own two products along: dW2 (+ db2), dW1 (+ db1), dx and the scalar of the loss leave as ONE launch
    (sgemm_triple_wgrad2_xgrad).  Nothing else is collected in between.  The held product reads `keep` and writes `touched`
    later: an in-place writer into any of them (flush_lazy_readers), the end of the backward pass and everything inside the
    library that shows results to the host launch it first."""
    held = False
    touched = set()       # id(HipBuffer) of what the held product reads or writes
    keep = ()

    @staticmethod
    def hold(reads, writes):
        H = HeldPair
        H.keep = tuple(t for t in reads + writes if t is not None)
        H.touched = set(id(t._data) for t in H.keep)
        _l.check(_l.lib().lg_gemm_pair_hold())
        H.held = True

    @staticmethod
    def resume() -> bool:
        """the caller collects into the open bracket and ends it (lg_gemm_pair_end) itself, then calls done()"""
        if not HeldPair.held:
            return False
        _l.check(_l.lib().lg_gemm_pair_resume())
        return True

    @staticmethod
    def done():
        H = HeldPair
        H.held, H.keep = False, ()
        H.touched.clear()

    @staticmethod
    def release():
        """launch what is held now (a launch of its own)"""
        if HeldPair.held:
            HeldPair.done()
            _l.check(_l.lib().lg_gemm_pair_end())


def _pass_ends():
    HeldPair.release()
    GradGroup.pass_ends()


def flush_lazy_readers(t) -> None:
    """call before any kernel WRITES into storage that already exists (in-place operators, fill, setitem, uploads,
    accumulating epilogues, optimizer updates, collectives): lazy tensors that were defined from the block's current
    contents (`relu` of a dense tensor, see HipTensor._lazy_source) are computed first, so that - like the reference,
    which evaluates relu at once (cpu/ops.py:226) - a later in-place change of the source never shows in them.
    Costs one attribute test when nobody is waiting (the normal case)."""
    if GradGroup.touched and not GradGroup.issuing and id(t._data) in GradGroup.touched:
        GradGroup.flush()              # storage that a queued launch will read or write
    if HeldPair.held and id(t._data) in HeldPair.touched:
        HeldPair.release()
    _make_lazy_readers_real(t)
    buf = t._data
    if buf is not None and buf.frozen:
        raise RuntimeError("in-place write into a frozen HipTensor (HipTensor.freeze(): a constant whose derived copies hipGraphs may hold)")
    if buf is not None and buf.derived is not None:
        buf.derived = None             # cached results made from the old contents (ids tiled once, then refreshed in place)


def _make_lazy_readers_real(t) -> None:
    buf = t._data
    if buf is not None and buf.lazy_readers:
        waiting, buf.lazy_readers = buf.lazy_readers, None
        for ref in waiting:
            reader = ref()
            if reader is not None and reader._data is None:
                reader._materialize()


class BackwardUpdate(object):
    """The optimizer's update applied by the kernels that MAKE the gradients (include/lghip.h: lg_adam_plan_* /
    lg_adam_epilogue_*): the weight-gradient GEMM's epilogue, its row-sum column (bias gradients), the slab workgroups of the
    skinny-head backward.  The optimizer's own launch disappears from the step (MNIST MLP: 4 launches -> 3; 5 -> 4 before the backward pass became one launch).

    The new parameter values cannot overwrite the old ones - `dx = g @ W` runs in the same launch as `dW = g^T @ x` - so the
    parameters live in TWO flat buckets and every step reads one and writes the other; the parameter tensors are re-pointed
    after each step (views handed out earlier keep the old values: take them per step, as the tape does).  The step number
    behind the bias corrections alternates between two device words the same way.  A hipGraph must therefore record an EVEN
    number of steps.  Gradients any other kernel produces (or that are added to, not overwritten) are applied by
    `finish()` in one extra launch - the result is the same either way, bit for bit: the same per-element arithmetic."""

    def __init__(self, flat_p, parameters, grad, m, v, offsets, lr, b1, b2, eps, grad_scale, belief, steps_done):
        lib = _l.lib()
        self.parameters, self.grad, self.offsets = tuple(parameters), grad, tuple(offsets)
        self.buckets = (flat_p, HipTensor.empty(flat_p._shape, requires_grad=False))
        self.m, self.v = m, v
        self.steps = HipTensor.from_numpy(np.asarray([steps_done, steps_done], dtype=np.int64), requires_grad=False)
        self.parity, self.armed, self._captured_steps = 0, False, 0
        self.plans = ([], [])
        n_params = len(self.parameters)
        for direction in (0, 1):
            src, dst = self.buckets[direction], self.buckets[1 - direction]
            for i, (a, b) in enumerate(zip(self.offsets[:-1], self.offsets[1:])):
                if b == a:
                    self.plans[direction].append(None)
                    continue
                plan = ctypes.c_void_p()
                _l.check(lib.lg_adam_plan_create(ctypes.byref(plan), src.ptr + 4 * a, dst.ptr + 4 * a, m.ptr + 4 * a, v.ptr + 4 * a, b - a,
                                                 self.steps.ptr + 8 * direction,
                                                 (self.steps.ptr + 8 * (1 - direction)) if i == self._first_nonempty() else None,
                                                 n_params, i + 1, lr, b1, b2, eps, grad_scale, 1 if belief else 0))
                self.plans[direction].append(plan.value)

        import weakref
        from .graph import HipGraph
        me = weakref.ref(self)

        def capture_ended():
            this = me()
            if this is None:
                HipGraph.capture_end_hooks.remove(capture_ended)
                return
            recorded, this._captured_steps = this._captured_steps, 0
            if recorded % 2:
                for _ in range(1):                    # undo the host-side flip the recorded (not executed) odd step left behind
                    this.parity ^= 1
                    for p in this.parameters:
                        p._data = this.buckets[this.parity]._data
                raise RuntimeError("a hipGraph recorded %d steps of an optimizer whose update rides in the backward kernels: its two "
                                   "parameter buckets alternate, record an EVEN number of steps per graph" % recorded)
        HipGraph.capture_end_hooks.append(capture_ended)

    def _first_nonempty(self) -> int:
        return next(i for i, (a, b) in enumerate(zip(self.offsets[:-1], self.offsets[1:])) if b > a)

    def arm(self) -> None:
        """before backward (optimizer.zero_grad): the kernels that overwrite a parameter's gradient apply its update"""
        if self.armed:
            raise RuntimeError("the optimizer's update rides in the backward kernels: zero_grad() must be followed by backward() and "
                               "step() before the next zero_grad()")
        flush_lazy_readers(self.buckets[1 - self.parity])      # anything still waiting to read the bucket about to be overwritten
        lib = _l.lib()
        for plan, a, b in zip(self.plans[self.parity], self.offsets[:-1], self.offsets[1:]):
            if plan is not None:
                _l.check(lib.lg_adam_epilogue_arm(self.grad.ptr + 4 * a, b - a, plan))
        self.armed = True

    def finish(self) -> tuple:
        """after backward (optimizer.step): apply what no kernel took, re-point the parameters at the bucket just written.
        Returns (updates applied by backward kernels, updates applied here)."""
        assert self.armed, "optimizer.step() without zero_grad() + backward() before it"
        taken, here = ctypes.c_int(0), ctypes.c_int(0)
        _l.check(_l.lib().lg_adam_epilogue_finish(ctypes.byref(taken), ctypes.byref(here)))
        from .graph import HipGraph
        if HipGraph.capturing:
            self._captured_steps += 1
        self.armed = False
        self.parity ^= 1
        now = self.buckets[self.parity]._data
        for p in self.parameters:
            p._data = now
        return taken.value, here.value

    def current_bucket(self):
        return self.buckets[self.parity]

    def disarm(self) -> None:
        """after a backward pass that failed: forget the armed updates (some may have been applied: the parameters are then
        not to be trusted - this only makes the library usable again)"""
        if self.armed:
            _l.check(_l.lib().lg_adam_epilogue_disarm())
            self.armed = False

    def __del__(self):
        if _l._lib is None:
            return
        if self.armed:
            _l._lib.lg_adam_epilogue_disarm()
        for direction in (0, 1):
            for plan in self.plans[direction]:
                if plan is not None:
                    _l._lib.lg_adam_plan_destroy(plan)


class HipDevice(object):
    """The GPU this process is bound to (one process per GPU; reference analog: OpenCLDevice,
    opencl/device.py:68-115 - context + in-order queue + memory pool)."""

    @staticmethod
    def is_available() -> bool:
        try:
            _l.lib()
            return True
        except (_l.HipError, OSError):
            return False

    @staticmethod
    def info() -> dict:
        di = _l.DeviceInfo()
        _l.check(_l.lib().lg_device_info(ctypes.byref(di)))
        return {"name": di.name.decode(), "arch": di.arch.decode(), "compute_units": di.compute_units,
                "clock_mhz": di.clock_mhz, "wavefront_size": di.wavefront_size,
                "lds_bytes_per_cu": di.lds_bytes_per_cu, "hbm_bytes": di.hbm_bytes, "l2_bytes": di.l2_bytes}

    @staticmethod
    def synchronize() -> None:
        _l.check(_l.lib().lg_sync())

    @staticmethod
    def pool_stats() -> dict:
        r, u, n = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _l.check(_l.lib().lg_pool_stats(ctypes.byref(r), ctypes.byref(u), ctypes.byref(n)))
        return {"reserved_bytes": r.value, "in_use_bytes": u.value, "hip_malloc_calls": n.value}

    @staticmethod
    def trim_pool() -> None:
        _l.check(_l.lib().lg_pool_trim())


class HipTensor(AbstractTensor):

    _adopt_first_grad = True      # intermediates share their first gradient (see AbstractTensor)

    @staticmethod
    def _backward_pass_begins(n_nodes):
        GradGroup.pass_begins(n_nodes)
        return _pass_ends

    def __init__(self, buffer: HipBuffer, shape: tuple, strides: tuple = None, offset: int = 0,
                 dtype: type = np.float32, requires_grad: bool = True):
        assert buffer is None or buffer.__class__ is HipBuffer        # None: a lazy tensor, see _lazy_source
        # (AbstractTensor.__init__ inlined: this constructor runs ~17 times per eager MLP step)
        self._data, self._grad, self._grad_shared, self._requires_grad, self._ctx = buffer, None, False, requires_grad, None
        self._dtype = dtype if dtype.__class__ is np.dtype else np.dtype(dtype)
        if shape.__class__ is not tuple or (shape and shape[0].__class__ is not int):
            shape = tuple(map(int, shape))      # lists, numpy integers
        self._shape = shape
        if strides is None:
            self._strides, self._dense = contiguous_strides(shape), True
        else:
            self._strides, self._dense = tuple(map(int, strides)), None      # None: not looked at yet (is_contiguous)
            assert len(shape) == len(self._strides), "Shapes and strides do not align! (%s <-> %s)" % (shape, self._strides)
        self._offset = offset if offset.__class__ is int else int(offset)
        self._byte_offset = self._offset * self._dtype.itemsize
        assert len(shape) <= 8, "HipTensor supports at most 8 dimensions"

    @property
    def dtype(self):
        return self._dtype

    @property
    def shape(self) -> tuple:
        return self._shape

    @property
    def strides(self) -> tuple:
        return self._strides

    @property
    def offset(self) -> int:
        return self._offset

    # A LAZY tensor has no buffer yet: `_lazy_source` says how to make it.
    #   ("relu", t)                      created by relu.forward; a consumer that can fold the relu into its own kernel
    #                                    (linear, the head kernels) reads t instead and the relu never runs
    #   ("head", x, relu, weight, bias)  created by linear.forward for a skinny output layer (<= 16 features):
    #                                    act(x) @ weight^T + bias, act = relu if `relu` else identity; loss.mse computes
    #                                    it together with the loss in one launch (csrc/head.hip)
    #   ("linear", x, weight)            created by linear.forward for a large bias-free product x @ weight^T: `+ bias` right
    #                                    after it becomes the GEMM's bias epilogue (ops._add_folding_bias)
    #   ("mse_rows", row_loss, n)        the scalar loss of that launch: (sum(row_loss) * (1/n)) * 0.5.  The backward
    #                                    launch of the head finishes it with a spare workgroup; reading it earlier costs
    #                                    one small launch (lg_mse_finalize_f32)
    # Anything else that asks for `data` / `ptr` runs the plain kernel first - so a lazy tensor behaves like any other
    # everywhere, it just costs nothing until someone looks.  In-place writers into the sources' storage compute the
    # waiting tensor first (flush_lazy_readers), so it is a snapshot like the reference's eager result.
    _lazy_source = None

    # set on a gradient tensor by the kernel that produced it when the same launch already wrote relu.backward's result
    # for it: (t, out_grad * (t >= 0)) - relu.backward of exactly that t then returns the second tensor without a launch
    _relu_bwd_done = None

    # set on the `err` tensor of a fused head + mse forward: weak reference to the loss tensor whose scalar is not finished
    # yet (`("mse_rows", ...)` above) - the head's backward launch, which receives err as its gradient, finishes it
    _unfinished_loss = None

    # set on the `err` tensor of a fused head + mse forward that also wrote relu.backward's result for the case that backward()
    # starts at this loss: (pre, weight, dx, gpre, token) - valid while `token` is still registered with the storage of pre, weight and
    # err (HipBuffer.derived["head_grad"]: any in-place writer drops it), see ops.head_mse_forward / ops._head_backward
    _head_grad_ahead = None

    @property
    def data(self):
        if self._data is None:
            self._materialize()
        return self._data

    def is_lazy(self) -> bool:
        return self._data is None

    def _materialize(self) -> None:
        from . import ops as _ops
        kind = self._lazy_source[0]
        if kind == "relu":
            src = self._lazy_source[1]
            out = HipTensor.empty(self._shape, requires_grad=False)
            _ops._ew(_l.EW_RELU, self._shape, [src], out=out)
        elif kind == "mse_rows":
            _, row_loss, n = self._lazy_source
            out = HipTensor.empty((), requires_grad=False)
            _l.check(_l.lib().lg_mse_finalize_f32(row_loss.ptr, row_loss.numel(), n, out.ptr))
        elif kind == "linear":
            _, x, weight = self._lazy_source
            out = _ops._gemm(x, _ops._swap_last(weight))

        else:
            assert kind == "head"
            _, x, relu, weight, bias = self._lazy_source
            if relu:
                out = _ops._gemm_fused(x, _ops._swap_last(weight), bias=bias, relu_a=True)[0]
            else:
                out = _ops._gemm(x, _ops._swap_last(weight), bias=bias)
        self._data, self._offset, self._byte_offset, self._lazy_source = out._data, out._offset, out._byte_offset, None

    def freeze(self) -> "HipTensor":
        """declare this tensor's storage a CONSTANT (a model's position ids): every later in-place write into it raises, and results
        the library derives from its contents and keeps (the ids tiled over a batch) may be recorded in hipGraphs without being made
        anew at every capture"""
        self.data.frozen = True
        return self

    def _watch_sources(self, *sources) -> None:
        """register this lazy tensor with the storage of everything it will read (see flush_lazy_readers)"""
        import weakref
        ref = weakref.ref(self)
        for t in sources:
            if t is None:
                continue
            buf = t.data
            if buf.lazy_readers is None:
                buf.lazy_readers = []
            elif len(buf.lazy_readers) >= 64:        # storage nobody ever writes (relu of a constant, step after step)
                buf.lazy_readers = [r for r in buf.lazy_readers if r() is not None and r()._data is None]
            buf.lazy_readers.append(ref)

    @property
    def ptr(self) -> int:
        """device address of element [0, ..., 0]"""
        buf = self._data
        if buf is None:
            buf = self.data                  # a lazy tensor: computed now
        return buf.ptr + self._byte_offset

    def numel(self) -> int:
        n = 1
        for s in self._shape:
            n *= s
        return n

    """ Initializers """

    @staticmethod
    def empty(shape, dtype: type = np.float32, requires_grad: bool = True) -> "HipTensor":
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        dtype = np.dtype(dtype)
        n = 1
        for s in shape:
            n *= s
        return HipTensor(HipBuffer(n * dtype.itemsize), shape=shape, dtype=dtype, requires_grad=requires_grad)

    @staticmethod
    def zeros(shape, dtype: type = np.float32, requires_grad: bool = True) -> "HipTensor":
        return HipTensor.empty(shape, dtype, requires_grad).fill(0).detach()

    @staticmethod
    def ones(shape, dtype: type = np.float32, requires_grad: bool = True) -> "HipTensor":
        return HipTensor.empty(shape, dtype, requires_grad).fill(1).detach()

    @staticmethod
    def uniform(low, high, shape, dtype: type = np.float32, requires_grad: bool = True) -> "HipTensor":
        # drawn on the host exactly like CpuTensor.uniform (float64 draw, then cast: cpu/tensor.py:36-37)
        # so that both backends start from bit-identical values for the same numpy seed
        a = np.random.uniform(low, high, size=shape).astype(dtype)
        return HipTensor.from_numpy(a, requires_grad=requires_grad)

    @classmethod
    def xavier(cls, shape, requires_grad: bool = True) -> "HipTensor":
        # U(-1,1)/sqrt(numel) (tensor.py:85-89) formed on the host exactly as the CPU backend does - numpy divides the
        # fp32 draw by the float64 sqrt in double and rounds once - so both backends start from identical bits
        a = np.random.uniform(-1, 1, size=shape).astype(np.float32)
        a /= np.sqrt(a.size)
        return HipTensor.from_numpy(a, requires_grad=requires_grad)

    @staticmethod
    def from_numpy(a: np.ndarray, requires_grad: bool = True) -> "HipTensor":
        a = np.asarray(a)
        if not a.flags["C_CONTIGUOUS"]:
            a = a.copy(order="C")          # (np.ascontiguousarray would turn a 0-d array into shape (1,))
        t = HipTensor.empty(a.shape, dtype=a.dtype, requires_grad=requires_grad)
        if a.nbytes > 0:
            _l.check(_l.lib().lg_memcpy_h2d(t.ptr, a.ctypes.data, a.nbytes))
        return t

    # set by `reshape` on a dense view of a leaf tensor: the leaf whose gradient buffer this view's gradient lands in
    _view_of_leaf = None

    def _grad_accumulator(self):
        """AbstractTensor's rule (a leaf that already owns a dense gradient), extended to dense reshape views of such a
        leaf: `leaf.grad += g.reshape(leaf.shape)` - what reshape.backward + add_grad would do with the view's gradient -
        is the same as adding g into a reshaped view of leaf.grad, so a backward kernel may accumulate there directly
        (and report None for the view: the reshape node then has nothing left to propagate)."""
        acc = AbstractTensor._grad_accumulator(self)
        if acc is None and self._view_of_leaf is not None and self._grad is None:
            base = AbstractTensor._grad_accumulator(self._view_of_leaf)
            if base is not None and base.is_contiguous() and base.numel() == self.numel():
                return HipTensor(base.data, self._shape, None, base._offset, base._dtype, requires_grad=False)
        return acc

    def _consume_zero_pending(self) -> bool:
        if self._view_of_leaf is not None and self._grad is None:
            return self._view_of_leaf._consume_zero_pending()
        return AbstractTensor._consume_zero_pending(self)

    _unit_seeds = {}          # shape -> constant tensor of ones used as the backward seed of item tensors

    def _seed_gradient(self):
        # the seed of a loss is always the same one-element tensor: keep ONE read-only constant per shape instead of
        # an allocation + fill kernel per backward pass (shared=True: the tape never writes into it)
        if self._shape in ((), (1,)) and self._dtype == np.float32:
            seed = HipTensor._unit_seeds.get(self._shape)
            if seed is None:
                from .graph import HipGraph
                if not HipGraph.capturing:      # while capturing, a fill would only be recorded, not executed
                    seed = HipTensor._unit_seeds[self._shape] = HipTensor.ones(self._shape, requires_grad=False)
                    seed._is_unit_constant = True       # lets a loss skip the multiplication by its own seed
            if seed is not None:
                return seed, True
        return HipTensor.ones(self._shape, dtype=self._dtype, requires_grad=False), False

    """ Data movement """

    def upload_(self, a: np.ndarray) -> "HipTensor":
        """overwrite this (dense) tensor with a host array of the same shape and dtype WITHOUT waiting: the data is
        staged in pinned memory and copied by a stream-ordered DMA.  This is how a training loop refreshes the static
        input tensors of a captured graph (autograd/hip/graph.py) between two replays."""
        a = np.asarray(a)
        assert self.is_contiguous() and a.shape == self._shape and a.dtype == self._dtype, \
            "upload_: need a dense tensor and an array of shape %s / dtype %s" % (self._shape, self._dtype)
        if not a.flags["C_CONTIGUOUS"]:
            a = a.copy(order="C")
        if a.nbytes > 0:
            flush_lazy_readers(self)
            _l.check(_l.lib().lg_memcpy_h2d_async(self.ptr, a.ctypes.data, a.nbytes))
        return self

    def is_contiguous(self) -> bool:
        dense = self._dense                  # shape and strides of a tensor object never change after construction ...
        if dense is None:
            dense, expect = True, 1
            for s, st in zip(reversed(self._shape), reversed(self._strides)):
                if s != 1 and st != expect:
                    dense = False
                    break
                expect *= s
            self._dense = dense
        return dense

    def contiguous(self) -> "HipTensor":
        """self if already dense row-major, else a gathered copy (strided copy kernel)"""
        if self.is_contiguous():
            return self
        out = HipTensor.empty(self._shape, dtype=self._dtype, requires_grad=self.requires_grad)
        nd = len(self._shape)
        _l.check(_l.lib().lg_copy_strided(self._dtype.itemsize, nd, _l.i64(self._shape), out.ptr, _l.i64(out._strides),
                                          self.ptr, _l.i64(self._strides)))
        return out

    def _is_dense_permutation(self) -> bool:
        """True if the elements occupy one gap-free run of memory in SOME dimension order (e.g. a transposed
        view of a dense tensor)"""
        dims = sorted(((st, s) for s, st in zip(self._shape, self._strides) if s != 1), reverse=True)
        expect = 1
        for st, s in reversed(dims):
            if st != expect:
                return False
            expect *= s
        return True

    def copy(self, requires_grad: bool = True) -> "HipTensor":
        """independent copy; a dense-but-permuted view keeps its memory layout (like numpy's order='K'), so
        the copy is one flat device memcpy and a later transpose back is dense again"""
        nd = len(self._shape)
        if nd > 1 and not self.is_contiguous() and self._is_dense_permutation():
            out = HipTensor(HipBuffer(self.numel() * self._dtype.itemsize), self._shape, self._strides, 0, self._dtype,
                            requires_grad=requires_grad)
            _l.check(_l.lib().lg_memcpy_d2d(out.ptr, self.ptr, self.numel() * self._dtype.itemsize))
            return out
        out = HipTensor.empty(self._shape, dtype=self._dtype, requires_grad=requires_grad)
        _l.check(_l.lib().lg_copy_strided(self._dtype.itemsize, nd, _l.i64(self._shape), out.ptr, _l.i64(out._strides),
                                          self.ptr, _l.i64(self._strides)))
        return out

    def numpy(self) -> np.ndarray:
        """blocking device-to-host copy (the only implicit synchronisation point)"""
        src = self.contiguous()
        arr = np.empty(self._shape, dtype=self._dtype)
        if arr.nbytes > 0:
            _l.check(_l.lib().lg_memcpy_d2h(arr.ctypes.data, src.ptr, arr.nbytes))
        return arr

    def _fused_adam_step(self, grad, m, v, lr, b1, b2, eps, inv_bias1, inv_bias2, grad_scale, belief):
        """optional optimizer hook (optim.Adam(fused=True)): one kernel instead of ~14 elementwise launches"""
        from .ops import adam_step_
        flush_lazy_readers(self)
        adam_step_(self, grad, m, v, lr, b1, b2, eps, inv_bias1, inv_bias2, grad_scale, belief)

    def _fused_mse(self, y_hat):
        """optional loss hook (loss.mse): (loss, err) from one kernel instead of seven tape ops; a still-lazy skinny
        output layer is computed in the same launch (csrc/head.hip)"""
        from .ops import mse_forward, head_mse_forward
        if self._data is None and self._lazy_source is not None and self._lazy_source[0] == "head":
            fused = head_mse_forward(self, y_hat)
            if fused is not None:
                return fused
        return mse_forward(self, y_hat)

    def _fused_cross_entropy(self, labels):
        """optional loss hook (loss.cross_entropy): (loss, (softmax - onehot) / N) from one row-wise kernel"""
        from .ops import cross_entropy_forward
        return cross_entropy_forward(self, labels)

    def _fused_adam_multi_dev(self, grad, m, v, offsets, lr, b1, b2, eps, step_counter, grad_scale, belief):
        """self/grad/m/v are flat buckets holding len(offsets)-1 parameters: ONE launch updates them all and advances the step
        counter (`_new_step_counter(step, slots=...)`: every workgroup keeps a copy of its own - csrc/optim.hip)"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        assert offsets[-1] == self.numel()
        flush_lazy_readers(self)
        _l.check(_l.lib().lg_adam_multi_dev_f32(self.ptr, grad.ptr, m.ptr, v.ptr, len(offsets) - 1, _l.i64(tuple(offsets)),
                                                lr, b1, b2, eps, step_counter.ptr, step_counter.numel() - 2, grad_scale,
                                                1 if belief else 0))

    def _fused_adam_multi_p2p(self, grad, m, v, offsets, lr, b1, b2, eps, step_counter, grad_scale, belief):
        """`_fused_adam_multi_dev` of a data-parallel rank: the SAME launch first sums `grad` over the ranks through the peer
        windows (include/lghip_p2p.h) and also advances the step counter (`_new_step_counter(step, slots=...)`: every
        workgroup keeps a copy of its own, no carrier kernel needed)"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        assert offsets[-1] == self.numel()
        flush_lazy_readers(self)
        flush_lazy_readers(grad)
        _l.check(_l.lib().lg_p2p_adam_multi_dev_f32(self.ptr, grad.ptr, m.ptr, v.ptr, len(offsets) - 1, _l.i64(tuple(offsets)),
                                                    lr, b1, b2, eps, step_counter.ptr, step_counter.numel() - 2, grad_scale,
                                                    1 if belief else 0))

    def _new_backward_update(self, parameters, grad, m, v, offsets, lr, b1, b2, eps, grad_scale, belief, steps_done=0):
        """optional optimizer hook (optim.Adam.fuse_update_into_backward): self/grad/m/v are flat buckets; returns the object
        that arms / finishes the update-in-the-backward-kernels of every step (BackwardUpdate)"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        assert offsets[-1] == self.numel()
        return BackwardUpdate(self, parameters, grad, m, v, offsets, lr, b1, b2, eps, grad_scale, belief, steps_done)

    @staticmethod
    def _new_step_counter(step: int, slots: int = 0) -> "HipTensor":
        """device-resident optimizer step number for graph-captured training steps: int64[2 + slots] = (step, unused, then
        `slots` private copies of the step for the workgroups of `_fused_adam_multi_dev` / `_fused_adam_multi_p2p`)"""
        return HipTensor.from_numpy(np.asarray([step, 0] + [step] * slots, dtype=np.int64), requires_grad=False)

    def _fused_adam_step_dev(self, grad, m, v, lr, b1, b2, eps, step_counter, t_mul, t_add, grad_scale, belief):
        """like `_fused_adam_step`, but t = step_counter * t_mul + t_add is evaluated on the device"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        flush_lazy_readers(self)
        _l.check(_l.lib().lg_adam_step_dev_f32(self.ptr, grad.ptr, m.ptr, v.ptr, self.numel(), lr, b1, b2, eps,
                                               step_counter.ptr, t_mul, t_add, grad_scale, 1 if belief else 0))

    @staticmethod
    def _advance_step_counter(step_counter, delta: int = 1) -> None:
        """one tiny launch: for optimizers whose update kernels only READ the counter (the per-parameter `_fused_adam_step_dev`)"""
        _l.check(_l.lib().lg_counter_add_i64(step_counter.ptr, delta))

    def __repr__(self):
        return "HipTensor(shape=%s, strides=%s, dtype=%s)" % (self._shape, self._strides, self._dtype)


# registers all hip ops (bottom import: ops needs HipTensor)
from . import ops  # noqa: E402,F401
