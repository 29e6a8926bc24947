"""hipGraph capture of the library's stream: record a launch-bound step once, replay it with ONE host
call (MI355X-first replacement for a tracing compiler; the reference blocks after every kernel,
opencl/kernels.py:194).

    step()                                   # warm up eagerly: pool, optimizer state, kernels are loaded
    g = HipGraph()
    with g.capture():
        loss = step()                        # the python tape runs once; kernels are recorded, not executed
    for _ in range(n):
        g.replay()                           # ~45 kernels, one hipGraphLaunch
    loss.item()                              # static output tensor: holds the last replay's value

Rules inside `capture()`: no host transfers or synchronisation (`numpy()`, `item()`, `from_numpy`), inputs
must already be device tensors (refresh them between replays with `t[...] = new` outside the graph), and any
value that changes per step must live in device memory (see `optim.Adam(device_step=True)`).  Memory the
captured kernels touch stays pinned to the graph until it is destroyed (lghip.h: lg_graph_*).
"""
import ctypes
from contextlib import contextmanager
from . import lib as _l


class HipGraph(object):

    capturing = False         # True inside a `capture()` block (kernels are being recorded, not executed)
    # callables run when a capture ends successfully (before the graph is instantiated): state that alternates per step on the
    # host - tensor.BackwardUpdate's two parameter buckets - checks that the recorded steps bring it back to where it started
    capture_end_hooks = []

    def __init__(self):
        self._exec = None

    @contextmanager
    def capture(self):
        assert self._exec is None, "HipGraph already holds a captured graph"
        L = _l.lib()
        _l.check(L.lg_graph_begin())
        HipGraph.capturing = True
        handle = ctypes.c_void_p()
        try:
            yield self
        except BaseException:
            HipGraph.capturing = False
            L.lg_graph_end(ctypes.byref(handle))     # leave capture mode, drop whatever was recorded
            if handle.value:
                L.lg_graph_destroy(handle)
            raise
        HipGraph.capturing = False
        _l.check(L.lg_graph_end(ctypes.byref(handle)))
        self._exec = handle
        try:
            for hook in list(HipGraph.capture_end_hooks):
                hook()
        except BaseException:
            self.destroy()
            raise

    def replay(self):
        assert self._exec is not None, "nothing captured"
        _l.check(_l._lib.lg_graph_launch(self._exec))

    def kernel_count(self) -> int:
        """kernel launches one replay performs"""
        assert self._exec is not None, "nothing captured"
        n = ctypes.c_int(0)
        _l.check(_l._lib.lg_graph_kernel_count(self._exec, ctypes.byref(n)))
        return n.value

    def destroy(self):
        if self._exec is not None and _l._lib is not None:
            _l._lib.lg_graph_destroy(self._exec)
        self._exec = None

    def __del__(self):
        self.destroy()


class GraphedStep(object):
    """A training (or inference) step recorded once and replayed: `step = GraphedStep(fn, optimizers=[opt])`.

    `fn()` must be a fixed-shape step over device tensors that performs no host transfer (see the rules in the
    module docstring) and returns the tensor(s) to read afterwards (e.g. the loss).  The first `warmup` calls run
    eagerly (they allocate optimizer state and fill the memory pool), the next call captures, every later call is
    one `hipGraphLaunch`.  Optimizers passed in must use `device_step=True`; their host-side step count is kept
    in line with the replays."""

    def __init__(self, fn, optimizers=(), warmup: int = 2):
        self._fn, self._opts, self._warmup = fn, tuple(optimizers), warmup
        for o in self._opts:
            assert getattr(o, "device_step", False), "GraphedStep needs optimizers created with device_step=True"
        self._graph, self._result, self._calls = None, None, 0

    def __call__(self):
        self._calls += 1
        if self._calls <= self._warmup:
            return self._fn()
        if self._graph is None:
            before = [o.t for o in self._opts]
            self._graph = HipGraph()
            with self._graph.capture():
                self._result = self._fn()
            for o, t in zip(self._opts, before):
                o.t = t                                   # the capture pass ran the python bookkeeping, not the kernels
        self._graph.replay()
        for o in self._opts:
            o.on_graph_replay()
        return self._result

    def destroy(self):
        if self._graph is not None:
            self._graph.destroy()
        self._graph = None
