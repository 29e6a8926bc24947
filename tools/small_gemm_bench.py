"""Time the small products of a tiny-BERT encoder layer (1024 rows, hidden 128, intermediate 512) through the C ABI - forward
(NT) and input-gradient (NN) forms, 50 back-to-back launches each.  Run once per LG_GEMM_TILE setting."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import lib as L              # noqa: E402

lib = L.lib()
rng = np.random.RandomState(0)
t = lambda *s: HipTensor.from_numpy(rng.uniform(-1, 1, s).astype(np.float32))   # noqa: E731
x128, x512 = t(1024, 128), t(1024, 512)
w128, w512a, w512b, bias128, bias512 = t(128, 128), t(512, 128), t(128, 512), t(128), t(512)
out = HipTensor.empty((1024 * 512,), requires_grad=False)


def event():
    e = ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e)))
    return e


def timed(fn, reps=50):
    fn()
    e0, e1 = event(), event()
    L.check(lib.lg_event_record(e0))
    for _ in range(reps):
        fn()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float()
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return 1e3 * ms.value / reps


cases = [
    ("fwd  q/k/v/out  (1024x128, K=128) NT + bias", lambda: L.check(lib.lg_gemm_bias_f32(0, 1, 1024, 128, 128, x128.ptr, 128, 0, w128.ptr, 128, 0, out.ptr, 128, 0, 1, bias128.ptr))),
    ("fwd  inter      (1024x512, K=128) NT + bias", lambda: L.check(lib.lg_gemm_bias_f32(0, 1, 1024, 512, 128, x128.ptr, 128, 0, w512a.ptr, 128, 0, out.ptr, 512, 0, 1, bias512.ptr))),
    ("fwd  output     (1024x128, K=512) NT + bias", lambda: L.check(lib.lg_gemm_bias_f32(0, 1, 1024, 128, 512, x512.ptr, 512, 0, w512b.ptr, 512, 0, out.ptr, 128, 0, 1, bias128.ptr))),
    ("dx   q/k/v/out  (1024x128, K=128) NN", lambda: L.check(lib.lg_gemm_f32(0, 0, 1024, 128, 128, x128.ptr, 128, 0, w128.ptr, 128, 0, out.ptr, 128, 0, 1, 0))),
    ("dx   inter      (1024x128, K=512) NN", lambda: L.check(lib.lg_gemm_f32(0, 0, 1024, 128, 512, x512.ptr, 512, 0, w512a.ptr, 128, 0, out.ptr, 128, 0, 1, 0))),
    ("dx   output     (1024x512, K=128) NN", lambda: L.check(lib.lg_gemm_f32(0, 0, 1024, 512, 128, x128.ptr, 128, 0, w512b.ptr, 512, 0, out.ptr, 512, 0, 1, 0))),
]
total = 0.0
for name, fn in cases:
    us = timed(fn)
    total += us
    print("%-48s %7.2f us" % (name, us))
print("LG_GEMM_TILE=%s   sum %.1f us" % (os.environ.get("LG_GEMM_TILE", "auto"), total))
