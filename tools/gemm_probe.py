"""Time one GEMM shape through the C ABI (HIP events, back-to-back launches): python tools/gemm_probe.py M N K [tA tB] ...
Several shapes may be given as M,N,K,tA,tB tokens.  Used with the LG_GEMM_TILE / LG_GEMM_SLICES knobs."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import lib as L              # noqa: E402

lib = L.lib()


def event():
    e = ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e)))
    return e


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = event(), event()
    L.check(lib.lg_event_record(e0))
    for _ in range(reps):
        fn()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float()
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return 1e3 * ms.value / reps


for tok in sys.argv[1:]:
    M, N, K, ta, tb = (int(v) for v in tok.split(","))
    a = HipTensor.empty((M * K,), requires_grad=False)
    b = HipTensor.empty((K * N,), requires_grad=False)
    c = HipTensor.empty((M * N,), requires_grad=False)
    a.fill(0.5)
    b.fill(0.25)
    lda, ldb = (M if ta else K), (K if tb else N)
    us = timed(lambda: L.check(lib.lg_gemm_f32(ta, tb, M, N, K, a.ptr, lda, 0, b.ptr, ldb, 0, c.ptr, N, 0, 1, 0)))
    print("M=%-5d N=%-5d K=%-5d %s%s  %8.2f us  %7.2f TFLOP/s" % (M, N, K, "T" if ta else "N", "T" if tb else "N", us, 2.0 * M * N * K / us / 1e6))
