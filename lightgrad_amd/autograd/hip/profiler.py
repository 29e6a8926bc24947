"""Device-time profiler for the HIP backend: the reference's per-op table (autograd/utils/profiler.py:5-40) measures
host wall time, which on an asynchronous backend is only the dispatch cost.  `HipProfiler` brackets every op with a
pair of HIP events on the library's stream and reports the DEVICE time between them next to the host time
("extend with hipEvent").  Events are resolved once, when the `with` block ends."""
import ctypes
from collections import defaultdict
from ..utils.profiler import Profiler
from . import lib as _l


class HipProfiler(Profiler):

    def __init__(self):
        Profiler.__init__(self)
        self._open, self._pairs = [], []
        self.device_ms = {False: defaultdict(float), True: defaultdict(float)}      # backward? -> op name -> ms

    @staticmethod
    def _event():
        e = ctypes.c_void_p()
        _l.check(_l.lib().lg_event_create(ctypes.byref(e)))
        _l.check(_l._lib.lg_event_record(e))
        return e

    def on_enter(self, name, backward):
        self._open.append(self._event())

    def on_exit(self, name, backward):
        self._pairs.append((name, backward, self._open.pop(), self._event()))

    def __exit__(self, *args):
        Profiler.__exit__(self, *args)
        ms = ctypes.c_float()
        for name, backward, e0, e1 in self._pairs:
            _l.check(_l._lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))      # synchronises on e1
            self.device_ms[backward][name] += ms.value
            _l._lib.lg_event_destroy(e0)
            _l._lib.lg_event_destroy(e1)
        self._pairs = []

    def print(self, topn=-1):
        rows = list(self.table().items())
        rows = rows[:topn] if topn > 0 else rows
        print(" Function       | fwd host s (n)      dev ms | bwd host s (n)      dev ms\n" + "-" * 78)
        for n, (ft, fc, bt, bc) in rows:
            print(" %-15s| %8.4f (%4i) %9.3f | %8.4f (%4i) %9.3f" % (n, ft, fc, self.device_ms[False][n], bt, bc, self.device_ms[True][n]))
