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

    def replay(self):
        assert self._exec is not None, "nothing captured"
        _l.check(_l._lib.lg_graph_launch(self._exec))

    def destroy(self):
        if self._exec is not None and _l._lib is not None:
            _l._lib.lg_graph_destroy(self._exec)
        self._exec = None

    def __del__(self):
        self.destroy()
