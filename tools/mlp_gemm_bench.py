"""Time the six GEMMs of one MNIST-MLP step (batch 1024, 784->512->10) through the C ABI, each launched
back to back (HIP events): prints microseconds per call.  Used with the LG_GEMM_* tuning knobs."""
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
x, w1, h, w2, g2, g1 = t(1024, 784), t(512, 784), t(1024, 512), t(10, 512), t(1024, 10), t(1024, 512)
out = HipTensor.empty((1024 * 1024,), requires_grad=False)


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


def gemm(ta, tb, M, N, K, A, lda, B, ldb):
    return lambda: L.check(lib.lg_gemm_f32(ta, tb, M, N, K, A.ptr, lda, 0, B.ptr, ldb, 0, out.ptr, N, 0, 1, 0))


cases = {
    "fwd1  x @ W1^T   (1024x512, K=784) NT": gemm(0, 1, 1024, 512, 784, x, 784, w1, 784),
    "fwd2  h @ W2^T   (1024x10,  K=512) NT": gemm(0, 1, 1024, 10, 512, h, 512, w2, 512),
    "dh    g2 @ W2    (1024x512, K=10)  NN": gemm(0, 0, 1024, 512, 10, g2, 10, w2, 512),
    "dW2   g2^T @ h   (10x512,  K=1024) TN": gemm(1, 0, 10, 512, 1024, g2, 10, h, 512),
    "dx    g1 @ W1    (1024x784, K=512) NN": gemm(0, 0, 1024, 784, 512, g1, 512, w1, 784),
    "dW1   g1^T @ x   (512x784, K=1024) TN": gemm(1, 0, 512, 784, 1024, g1, 512, x, 784),
}
total = 0.0
for name, fn in cases.items():
    us = timed(fn)
    total += us
    print("%-44s %7.2f us" % (name, us))
out2 = HipTensor.empty((1024 * 1024,), requires_grad=False)


def pair():
    # what linear.backward issues when no gradient hook is waiting on dW: both products in ONE launch (lg_gemm_pair_*)
    L.check(lib.lg_gemm_pair_begin())
    L.check(lib.lg_gemm_f32(1, 0, 512, 784, 1024, g1.ptr, 512, 0, x.ptr, 784, 0, out.ptr, 784, 0, 1, 0))
    L.check(lib.lg_gemm_f32(0, 0, 1024, 784, 512, g1.ptr, 512, 0, w1.ptr, 784, 0, out2.ptr, 784, 0, 1, 0))
    L.check(lib.lg_gemm_pair_end())


us_pair = timed(pair)
print("%-44s %7.2f us" % ("dW1 and dx as one launch (lg_gemm_pair_*)", us_pair))
print("the six products with dW1 and dx paired: %.1f us" % (total - timed(cases["dx    g1 @ W1    (1024x784, K=512) NN"]) - timed(cases["dW1   g1^T @ x   (512x784, K=1024) TN"]) + us_pair))
print("knobs: LG_GEMM_TILE=%s LG_GEMM_SLICES=%s   total %.1f us" % (os.environ.get("LG_GEMM_TILE", "auto"), os.environ.get("LG_GEMM_SLICES", "auto"), total))
