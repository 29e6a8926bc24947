"""A/B the SGEMM kernel in ONE process (interleaved rounds, HIP events on the library stream):
    python tools/gemm_bench.py [N=4096] [rounds=5]
prints TFLOP/s per layout, median and best over rounds."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import lib as L              # noqa: E402

lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.RandomState(0)
a = HipTensor.from_numpy(rng.uniform(-1, 1, (n, n)).astype(np.float32))
b = HipTensor.from_numpy(rng.uniform(-1, 1, (n, n)).astype(np.float32))
c = HipTensor.empty((n, n), requires_grad=False)


def event():
    e = ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e)))
    return e


def timed(ta, tb, reps=10):
    e0, e1 = event(), event()
    L.check(lib.lg_gemm_f32(ta, tb, n, n, n, a.ptr, n, 0, b.ptr, n, 0, c.ptr, n, 0, 1, 0))
    L.check(lib.lg_event_record(e0))
    for _ in range(reps):
        L.check(lib.lg_gemm_f32(ta, tb, n, n, n, a.ptr, n, 0, b.ptr, n, 0, c.ptr, n, 0, 1, 0))
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float()
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return 2 * n ** 3 / (ms.value / reps * 1e-3) / 1e12


res = {}
for r in range(rounds):
    for tag, (ta, tb) in {"NN": (0, 0), "NT": (0, 1), "TN": (1, 0), "TT": (1, 1)}.items():
        res.setdefault(tag, []).append(timed(ta, tb))
print("LG_GEMM_TILE=%s n=%d  " % (os.environ.get("LG_GEMM_TILE", "0"), n) +
      "  ".join("%s med %.1f best %.1f" % (k, float(np.median(v)), max(v)) for k, v in res.items()))
