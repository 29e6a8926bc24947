"""Per-op wall-clock profiler hooked into every Function call.

Same surface as the reference's `lightgrad/autograd/utils/profiler.py:5-40`
(`Profiler` context manager with `print(topn)`, `Tracker(name, backward)`),
plus `Profiler.table()` returning the numbers instead of printing them.

On an asynchronous backend (HipTensor) the wall time of an op is its *host*
dispatch cost, not kernel time; use `rocprofv3 --kernel-trace --stats` for
device time (see profiles/).
"""
from time import perf_counter
from collections import defaultdict


class Profiler(object):
    # all profilers currently inside a `with` block (process-global, single-threaded like the reference)
    _active_profilers = []

    def __init__(self):
        self._fwd_time, self._fwd_calls = defaultdict(float), defaultdict(int)
        self._bwd_time, self._bwd_calls = defaultdict(float), defaultdict(int)

    def update(self, name, time_delta, backward=False):
        if backward:
            self._bwd_time[name] += time_delta
            self._bwd_calls[name] += 1
        else:
            self._fwd_time[name] += time_delta
            self._fwd_calls[name] += 1

    # hooks around every outermost tracked call (a device profiler records stream events here)
    def on_enter(self, name, backward):
        pass

    def on_exit(self, name, backward):
        pass

    def __enter__(self, *args):
        Profiler._active_profilers.append(self)
        return self

    def __exit__(self, *args):
        Profiler._active_profilers.remove(self)

    def table(self):
        """{name: (fwd_seconds, fwd_calls, bwd_seconds, bwd_calls)} sorted by forward time."""
        names = sorted(set(self._fwd_time) | set(self._bwd_time), key=lambda n: -self._fwd_time[n])
        return {n: (self._fwd_time[n], self._fwd_calls[n], self._bwd_time[n], self._bwd_calls[n]) for n in names}

    def print(self, topn=-1):
        rows = list(self.table().items())
        rows = rows[:topn] if topn > 0 else rows
        print(" Function       |   forward      \t|   backward   \n" + "-" * 70)
        for n, (ft, fc, bt, bc) in rows:
            print(" %-15s| %8.4fs (%i)\t| %8.4fs (%i) " % (n, ft, fc, bt, bc))
        print("\n")


class Tracker(object):
    # nesting depth: only the outermost tracked call is charged (reference profiler.py:31-34)
    _depth = 0

    def __init__(self, name, backward=False):
        self._name, self._backward = name, backward
        self._charge = (Tracker._depth == 0)

    def __enter__(self, *args):
        Tracker._depth += 1
        if self._charge:
            for p in Profiler._active_profilers:
                p.on_enter(self._name, self._backward)
        self._start = perf_counter()

    def __exit__(self, *args):
        Tracker._depth = max(0, Tracker._depth - 1)
        if self._charge:
            dt = perf_counter() - self._start
            for p in Profiler._active_profilers:
                p.update(self._name, dt, self._backward)
                p.on_exit(self._name, self._backward)
