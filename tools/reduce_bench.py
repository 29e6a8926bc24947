"""time full / row / column reductions of a 16384 x 8192 fp32 tensor (HIP events), GB/s of input traffic"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor
from lightgrad_amd.autograd.hip import lib as L
lib = L.lib()
big = (16384, 8192)
p = HipTensor.empty(big, requires_grad=False); p.fill(0.5)
sh, st = L.i64(big), L.i64(p.strides)
def event():
    e = ctypes.c_void_p(); L.check(lib.lg_event_create(ctypes.byref(e))); return e
def timed(fn, reps=10):
    fn(); e0, e1 = event(), event()
    L.check(lib.lg_event_record(e0))
    for _ in range(reps): fn()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float(); L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms))); return ms.value / reps
outs = {3: HipTensor.empty(()), 1: HipTensor.empty((8192,)), 2: HipTensor.empty((16384,))}
res = []
for name, op, mask in [("sum all", 0, 3), ("max all", 1, 3), ("sum axis0", 0, 1), ("sum axis1", 0, 2), ("max axis1", 1, 2)]:
    ms = timed(lambda: L.check(lib.lg_reduce(op, 2, sh, p.ptr, st, mask, outs[mask].ptr)))
    res.append("%s %.0f" % (name, big[0] * big[1] * 4 / ms / 1e6))
print("LG_RED_BLOCKS=%s GB/s: " % os.environ.get("LG_RED_BLOCKS", "default") + "  ".join(res))
