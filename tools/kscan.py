import ctypes, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from lightgrad_amd import HipTensor
from lightgrad_amd.autograd.hip import lib as L
lib = L.lib()
rng = np.random.RandomState(0)
x = HipTensor.from_numpy(rng.uniform(-1, 1, (1024, 4096)).astype(np.float32))
w = HipTensor.from_numpy(rng.uniform(-1, 1, (512, 4096)).astype(np.float32))
out = HipTensor.empty((1024 * 1024,), requires_grad=False)
def event():
    e = ctypes.c_void_p(); L.check(lib.lg_event_create(ctypes.byref(e))); return e
def timed(fn, reps=50):
    fn(); e0, e1 = event(), event()
    L.check(lib.lg_event_record(e0))
    for _ in range(reps): fn()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float(); L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return 1e3 * ms.value / reps
for K in [32, 64, 128, 256, 512, 1024, 2048, 4096]:
    us = timed(lambda: L.check(lib.lg_gemm_f32(0, 1, 1024, 512, K, x.ptr, 4096, 0, w.ptr, 4096, 0, out.ptr, 512, 0, 1, 0)))
    print("M=1024 N=512 K=%-5d NT: %7.2f us  (%.1f TF)" % (K, us, 2*1024*512*K/us/1e6))
