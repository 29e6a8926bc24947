"""Time the GEMM shapes of tiny-BERT's decoder (1024 rows, hidden 128, vocabulary 30522) and of one encoder Linear through the C ABI
(HIP events, 20 back-to-back launches).  Run once per LG_GEMM_TILE / LG_GEMM_SLICES setting to compare tiles."""
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
V = 30522
h, w, g, bias = t(1024, 128), t(V, 128), t(1024, V), t(V)
out = HipTensor.empty((1024 * V,), requires_grad=False)
out2 = HipTensor.empty((V * 128,), requires_grad=False)
x128, w128, g128 = t(1024, 128), t(128, 128), t(1024, 128)


def event():
    e = ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e)))
    return e


def timed(fn, reps=20):
    fn()
    e0, e1 = event(), event()
    L.check(lib.lg_event_record(e0))
    for _ in range(reps):
        fn()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float()
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return 1e3 * ms.value / reps


def pair(f1, f2):
    def run():
        L.check(lib.lg_gemm_pair_begin())
        f1()
        f2()
        L.check(lib.lg_gemm_pair_end())
    return run


fwd = lambda: L.check(lib.lg_gemm_bias_f32(0, 1, 1024, V, 128, h.ptr, 128, 0, w.ptr, 128, 0, out.ptr, V, 0, 1, bias.ptr))      # noqa: E731
dw = lambda: L.check(lib.lg_gemm_f32(1, 0, V, 128, 1024, g.ptr, V, 0, h.ptr, 128, 0, out2.ptr, 128, 0, 1, 0))                    # noqa: E731
dx = lambda: L.check(lib.lg_gemm_f32(0, 0, 1024, 128, V, g.ptr, V, 0, w.ptr, 128, 0, out.ptr, 128, 0, 1, 0))                     # noqa: E731
dw_s = lambda: L.check(lib.lg_gemm_f32(1, 0, 128, 128, 1024, g128.ptr, 128, 0, x128.ptr, 128, 0, out2.ptr, 128, 0, 1, 0))        # noqa: E731
dx_s = lambda: L.check(lib.lg_gemm_f32(0, 0, 1024, 128, 128, g128.ptr, 128, 0, w128.ptr, 128, 0, out.ptr, 128, 0, 1, 0))         # noqa: E731
cases = [("decoder fwd  h @ W^T + b (1024x30522, K=128) NT", fwd, 2 * 1024 * V * 128),
         ("decoder dW   g^T @ h     (30522x128, K=1024) TN", dw, 2 * 1024 * V * 128),
         ("decoder dx   g @ W       (1024x128, K=30522) NN", dx, 2 * 1024 * V * 128),
         ("decoder dW and dx as one launch", pair(dw, dx), 4 * 1024 * V * 128),
         ("encoder dW   g^T @ x     (128x128, K=1024) TN", dw_s, 2 * 1024 * 128 * 128),
         ("encoder dx   g @ W       (1024x128, K=128) NN", dx_s, 2 * 1024 * 128 * 128),
         ("encoder dW and dx as one launch", pair(dw_s, dx_s), 4 * 1024 * 128 * 128)]
for name, fn, flop in cases:
    us = timed(fn)
    print("%-52s %8.2f us  %6.1f TFLOP/s" % (name, us, flop / us * 1e-6))
print("knobs: LG_GEMM_TILE=%s LG_GEMM_SLICES=%s" % (os.environ.get("LG_GEMM_TILE", "auto"), os.environ.get("LG_GEMM_SLICES", "auto")))
