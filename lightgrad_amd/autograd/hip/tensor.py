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
    __slots__ = ("ptr", "nbytes", "lazy_readers", "__weakref__")

    def __init__(self, nbytes):
        self.ptr = None
        self.lazy_readers = None      # weak references to lazy tensors that will still READ this block (see flush_lazy_readers)
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
    MIN_NODES = 48
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


class SideStream(object):
    """Parameter-gradient kernels on a second HIP stream (lg_side_* in include/lghip.h).

    During a backward pass over a DEEP tape (a transformer: >= MIN_NODES tape nodes) the kernels that write parameter
    gradients in place - dW / db of every Linear, LayerNorm's weight and bias gradients, embedding scatter-adds - do not sit
    on the critical path: only the activation gradients feed the next node.  They are enqueued inside `with SideStream.bracket(...)`
    and run next to the main chain (as parallel branches when the pass is captured into a hipGraph); the pass ends with a
    join.  Shallow tapes (the MLP) keep everything on one stream - there the optimizer waits for every gradient at once and
    one launch for dW and dx together is the better deal (lg_gemm_pair_*).

    Hazards and how they are covered:
      * a side kernel READS tensors made on the main stream: the bracket forks after everything enqueued so far, and lazy
        operands are made real before the fork;
      * it WRITES into gradient buffers: `written` remembers their storage until the join; an in-place writer on the main
        stream (flush_lazy_readers is its hook) joins first;
      * memory freed meanwhile is parked inside the library until the join."""
    MIN_NODES = 48
    # OFF unless asked for: measured on tiny-BERT (profiles/README.md r2) a hipGraph with these parallel branches replays in
    # 3.8 ms instead of 0.86 ms - every edge between branches costs ~40 us in ROCm 7.2's graph executor - and the eager
    # tape is host-bound either way.  GradGroup (above) gets the same work off the critical path inside ONE stream.
    enabled = os.environ.get("LIGHTGRAD_SIDE_STREAM", "0") == "1"
    depth = 0             # nested backward passes (WrapperFunction replays its inner tape with Gradients.backward)
    active = False        # the running pass uses the side stream
    in_bracket = False
    written = set()       # id(HipBuffer) of storage with un-joined side writes

    @staticmethod
    def pass_begins(n_nodes):
        S = SideStream
        if S.depth == 0:
            S.active = S.enabled and n_nodes >= S.MIN_NODES and not GradGroup.active
        S.depth += 1
        return S.pass_ends

    @staticmethod
    def pass_ends():
        S = SideStream
        S.depth -= 1
        if S.depth == 0:
            S.active = False
            S.join()

    @staticmethod
    def join():
        S = SideStream
        if S.written:
            S.written.clear()
            _l.check(_l.lib().lg_side_join())

    @staticmethod
    def usable_for(*params) -> bool:
        """may the gradients of these leaf parameters be written from the side stream?  Not when someone waits for the
        moment they are enqueued (dist.DataParallel starts its exchange from that hook, on its own stream)"""
        S = SideStream
        if not S.active or S.in_bracket:
            return False
        for p in params:
            if p is not None and p._grad_written_hook is not None:
                return False
        return True

    class bracket(object):
        __slots__ = ()

        def __init__(self, reads, writes):
            for t in reads:
                if t is not None:
                    t.data                      # lazy operands become real on the main stream, before the fork
            for t in writes:
                if t is not None:
                    _make_lazy_readers_real(t)  # snapshots of the old contents, also before the fork
                    SideStream.written.add(id(t._data))

        def __enter__(self):
            _l.check(_l.lib().lg_side_begin())
            SideStream.in_bracket = True

        def __exit__(self, *exc):
            SideStream.in_bracket = False
            _l.check(_l.lib().lg_side_end())


def flush_lazy_readers(t) -> None:
    """call before any kernel WRITES into storage that already exists (in-place operators, fill, setitem, uploads,
    accumulating epilogues, optimizer updates, collectives): lazy tensors that were defined from the block's current
    contents (`relu` of a dense tensor, see HipTensor._lazy_source) are computed first, so that - like the reference,
    which evaluates relu at once (cpu/ops.py:226) - a later in-place change of the source never shows in them.
    Costs one attribute test when nobody is waiting (the normal case)."""
    if SideStream.written and not SideStream.in_bracket and id(t._data) in SideStream.written:
        SideStream.join()              # a main-stream writer into storage the side stream is still writing
    if GradGroup.touched and not GradGroup.issuing and id(t._data) in GradGroup.touched:
        GradGroup.flush()              # ... or that a queued launch will read or write
    _make_lazy_readers_real(t)


def _make_lazy_readers_real(t) -> None:
    buf = t._data
    if buf is not None and buf.lazy_readers:
        waiting, buf.lazy_readers = buf.lazy_readers, None
        for ref in waiting:
            reader = ref()
            if reader is not None and reader._data is None:
                reader._materialize()


class PendingUpload(object):
    """a host array on its way to a device staging slot (HipTensor.prefetch); consumed by HipTensor.commit_"""
    __slots__ = ("slot", "shape", "dtype", "nbytes", "keepalive")

    def __init__(self, slot, shape, dtype, nbytes, keepalive):
        self.slot, self.shape, self.dtype, self.nbytes, self.keepalive = slot, tuple(shape), np.dtype(dtype), nbytes, keepalive


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
    def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
        """numpy array in pinned (page-locked) host memory: the DMA engine reads it in place, so `HipTensor.prefetch` /
        `upload_` of such an array involve no host-side staging copy.  Freed when the array (and its views) die."""
        import weakref
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr = ctypes.c_void_p()
        _l.check(_l.lib().lg_host_malloc(ctypes.byref(ptr), max(nbytes, 1)))
        raw = (ctypes.c_char * max(nbytes, 1)).from_address(ptr.value)
        weakref.finalize(raw, _l._lib.lg_host_free, ptr)
        return np.frombuffer(raw, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)

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
        side_ends = SideStream.pass_begins(n_nodes)

        def ends():
            GradGroup.pass_ends()
            side_ends()
        return ends

    def __init__(self, buffer: HipBuffer, shape: tuple, strides: tuple = None, offset: int = 0,
                 dtype: type = np.float32, requires_grad: bool = True):
        assert isinstance(buffer, HipBuffer) or buffer is None        # None: a lazy tensor, see _lazy_source
        AbstractTensor.__init__(self, data=buffer, requires_grad=requires_grad)
        self._dtype = dtype if dtype.__class__ is np.dtype else np.dtype(dtype)
        self._shape = tuple(map(int, shape))
        if strides is None:
            self._strides, self._dense = contiguous_strides(self._shape), True
        else:
            self._strides, self._dense = tuple(map(int, strides)), None      # None: not looked at yet (is_contiguous)
        self._offset = int(offset)
        assert len(self._shape) == len(self._strides), \
            "Shapes and strides do not align! (%s <-> %s)" % (self._shape, self._strides)
        assert len(self._shape) <= 8, "HipTensor supports at most 8 dimensions"

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
        self._data, self._offset, self._lazy_source = out._data, out._offset, None

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
        return self.data.ptr + self._offset * self._dtype.itemsize

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

    @staticmethod
    def prefetch(a: np.ndarray) -> "PendingUpload":
        """start copying a host array to the device on the COPY stream and return at once; `tensor.commit_(pending)`
        later makes the compute stream wait for that DMA and moves the data into `tensor`.  Prefetching batch i+1
        before replaying the step on batch i overlaps the PCIe transfer with compute (examples/mnist.py --graph)."""
        a = np.asarray(a)
        if not a.flags["C_CONTIGUOUS"]:
            a = a.copy(order="C")
        assert a.nbytes > 0
        slot = ctypes.c_int(-1)
        _l.check(_l.lib().lg_prefetch_h2d(a.ctypes.data, a.nbytes, ctypes.byref(slot)))
        return PendingUpload(slot.value, a.shape, a.dtype, a.nbytes, a)

    def commit_(self, pending: "PendingUpload") -> "HipTensor":
        assert self.is_contiguous() and pending.shape == self._shape and pending.dtype == self._dtype, \
            "commit_: need a dense tensor of shape %s / dtype %s" % (pending.shape, pending.dtype)
        assert pending.slot >= 0, "this prefetch has already been committed"
        flush_lazy_readers(self)
        _l.check(_l.lib().lg_prefetch_commit(pending.slot, self.ptr, pending.nbytes))
        pending.slot, pending.keepalive = -1, None
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
        """self/grad/m/v are flat buckets holding len(offsets)-1 parameters: ONE launch updates them all"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        assert offsets[-1] == self.numel()
        flush_lazy_readers(self)
        self._flush_step_counter(step_counter)
        _l.check(_l.lib().lg_adam_multi_dev_f32(self.ptr, grad.ptr, m.ptr, v.ptr, len(offsets) - 1, _l.i64(tuple(offsets)),
                                                lr, b1, b2, eps, step_counter.ptr, grad_scale, 1 if belief else 0, 0))
        # advance = 0: the ticket form (the last working workgroup increments the counter) costs one contended atomic per
        # working workgroup - measured 2 % slower per MLP step (~400 tickets) than a separate 1-thread launch.  Cheaper
        # than both: the NEXT step's loss kernel carries the increment (see _advance_step_counter)
        self._advance_step_counter(step_counter, defer=True)

    def _fused_adam_multi_p2p(self, grad, m, v, offsets, lr, b1, b2, eps, step_counter, grad_scale, belief):
        """`_fused_adam_multi_dev` of a data-parallel rank: the SAME launch first sums `grad` over the ranks through the peer
        windows (include/lghip_p2p.h) and also advances the step counter (`_new_step_counter(step, slots=...)`: every
        workgroup keeps a copy of its own, no carrier kernel needed)"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        assert offsets[-1] == self.numel()
        flush_lazy_readers(self)
        flush_lazy_readers(grad)
        self._flush_step_counter(step_counter)
        _l.check(_l.lib().lg_p2p_adam_multi_dev_f32(self.ptr, grad.ptr, m.ptr, v.ptr, len(offsets) - 1, _l.i64(tuple(offsets)),
                                                    lr, b1, b2, eps, step_counter.ptr, step_counter.numel() - 2, grad_scale,
                                                    1 if belief else 0))

    @staticmethod
    def _new_step_counter(step: int, slots: int = 0) -> "HipTensor":
        """device-resident optimizer step number for graph-captured training steps: int64[2] = (step, arrival ticket), plus
        `slots` private copies of the step for the workgroups of `_fused_adam_multi_p2p`"""
        return HipTensor.from_numpy(np.asarray([step, 0] + [step] * slots, dtype=np.int64), requires_grad=False)

    def _fused_adam_step_dev(self, grad, m, v, lr, b1, b2, eps, step_counter, t_mul, t_add, grad_scale, belief):
        """like `_fused_adam_step`, but t = step_counter * t_mul + t_add is evaluated on the device"""
        for t in (self, grad, m, v):
            assert t.is_contiguous() and t._shape == self._shape and t._dtype == np.float32
        flush_lazy_readers(self)
        self._flush_step_counter(step_counter)
        _l.check(_l.lib().lg_adam_step_dev_f32(self.ptr, grad.ptr, m.ptr, v.ptr, self.numel(), lr, b1, b2, eps,
                                               step_counter.ptr, t_mul, t_add, grad_scale, 1 if belief else 0))

    # Device step counters whose "+1" has not been enqueued yet.  An optimizer step ends by advancing its counter; that
    # increment only has to land before the optimizer's NEXT kernel reads the counter, so instead of a 1-thread launch of
    # its own (4 us of a 100 us training step) it waits here for a kernel that runs exactly once per training step anyway:
    # the fused loss of the next forward pass (lg_head_fwd_f32 / lg_mse_bump_f32 take the counter as an argument).  If no
    # such kernel came by, the optimizer flushes the increment itself before it reads the counter again.
    #
    # hipGraph capture (autograd/hip/graph.py calls _capture_begins / _capture_ended / _graph_replayed): a capture pass
    # executes nothing, so the list is restored when it ends, and what the recorded kernels do to it - in order - is
    # replayed with them:
    #   * a loss kernel recorded in the capture carried a waiting increment -> an optimizer recorded after it defers again
    #     (warm capture of one or several whole training steps: one increment per recorded step and no launch for it)
    #   * nothing carried it -> the optimizer records its own 1-thread increment in the graph, as it would eagerly
    #   * an increment of the step BEFORE the capture that no recorded kernel carried is executed once, when the capture
    #     has ended (it belongs to that step, not to every replay)
    _deferred_step_advances = []          # weak references: a counter dies with its optimizer
    _capture_state = None                 # during a capture: {"snapshot", "own", "events", "owed"}

    @staticmethod
    def _waiting_step_counters():
        """the live counters with an increment waiting, oldest first (entries of dead optimizers are dropped)"""
        alive = [(r, r()) for r in HipTensor._deferred_step_advances]
        HipTensor._deferred_step_advances[:] = [r for r, c in alive if c is not None]
        return [c for _, c in alive if c is not None]

    @staticmethod
    def _drop_waiting(step_counter):
        """remove the oldest waiting entry of this counter; returns the entry (a weakref) or None"""
        pending = HipTensor._deferred_step_advances
        for i, ref in enumerate(pending):
            if ref() is step_counter:
                del pending[i]
                return ref
        return None

    @staticmethod
    def _advance_step_counter(step_counter, delta: int = 1, defer: bool = False) -> None:
        if defer and delta == 1:
            import weakref
            cap = HipTensor._capture_state
            if cap is None or any(kind == "take" and ref() is step_counter for kind, ref in cap["events"]):
                entry = weakref.ref(step_counter)
                HipTensor._deferred_step_advances.append(entry)
                if cap is not None:
                    cap["own"].append(entry)
                    cap["events"].append(("defer", entry))
                return
        _l.check(_l.lib().lg_counter_add_i64(step_counter.ptr, delta))

    @staticmethod
    def _flush_step_counter(step_counter) -> None:
        """the optimizer is about to read `step_counter`: enqueue an increment that is still waiting for a carrier"""
        entry = HipTensor._drop_waiting(step_counter)
        if entry is None:
            return
        cap = HipTensor._capture_state
        if cap is not None and not any(entry is e for e in cap["own"]):
            cap["owed"].append(step_counter)              # belongs to the step before the capture: executed once, afterwards
            return
        if cap is not None:
            cap["events"].append(("take", entry))         # deferred inside this capture, settled inside it: recorded launch
        _l.check(_l.lib().lg_counter_add_i64(step_counter.ptr, 1))

    @staticmethod
    def _take_deferred_step_advance():
        """for a loss kernel of a TRAINING forward pass: the counter it should increment, or None"""
        from ..grads import Gradients
        # depth 1 = inside a first-class op's forward with gradients otherwise enabled; deeper = the user's no_grad()
        if HipTensor._deferred_step_advances and Gradients._disable_depth == 1:
            waiting = HipTensor._waiting_step_counters()
            if waiting:
                entry = HipTensor._deferred_step_advances.pop(0)
                if HipTensor._capture_state is not None:
                    HipTensor._capture_state["events"].append(("take", entry))
                return waiting[0]
        return None

    @staticmethod
    def _capture_begins() -> None:
        HipTensor._capture_state = {"snapshot": list(HipTensor._deferred_step_advances), "own": [], "events": [], "owed": []}

    @staticmethod
    def _capture_ended(ok: bool = True):
        """restore the waiting list (the capture pass executed nothing), settle what is owed to the step before the
        capture, and hand the graph what its recorded kernels do to the list per replay: [("take" | "defer", counter ref)]"""
        cap, HipTensor._capture_state = HipTensor._capture_state, None
        HipTensor._deferred_step_advances[:] = cap["snapshot"]
        if not ok:
            return []
        for counter in cap["owed"]:
            HipTensor._drop_waiting(counter)
            _l.check(_l.lib().lg_counter_add_i64(counter.ptr, 1))
        return cap["events"]

    @staticmethod
    def _graph_replayed(events) -> None:
        import weakref
        for kind, ref in events:
            c = ref()
            if c is None:
                continue
            if kind == "take":
                HipTensor._drop_waiting(c)
            else:
                HipTensor._deferred_step_advances.append(weakref.ref(c))

    def __repr__(self):
        return "HipTensor(shape=%s, strides=%s, dtype=%s)" % (self._shape, self._strides, self._dtype)


# registers all hip ops (bottom import: ops needs HipTensor)
from . import ops  # noqa: E402,F401
