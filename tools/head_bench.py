"""Time the two head kernels (csrc/head.hip) in isolation, back to back, at the MNIST shape; LG_HEAD_DBG switches parts
off (see the kernel arguments) to attribute the time:   python tools/head_bench.py [rows hidden outs]"""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                                    # noqa: E402
from lightgrad_amd.autograd.hip import lib as L                        # noqa: E402

lib = L.lib()
rows, hidden, outs = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 512, 10)
rng = np.random.RandomState(0)
t = lambda *s: HipTensor.from_numpy(rng.uniform(-1, 1, s).astype(np.float32), requires_grad=False)   # noqa: E731
x, w, b, tgt, g = t(rows, hidden), t(outs, hidden), t(outs), t(rows, outs), t(rows, outs)
y, err, loss, row_loss = HipTensor.empty((rows, outs)), HipTensor.empty((rows, outs)), HipTensor.empty(()), HipTensor.empty((rows,))
dx, gpre, dw, db = HipTensor.empty((rows, hidden)), HipTensor.empty((rows, hidden)), HipTensor.empty((outs, hidden)), HipTensor.empty((outs,))


def timed(fn, n=200):
    for _ in range(20):
        fn()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e0)))
    L.check(lib.lg_event_create(ctypes.byref(e1)))
    L.check(lib.lg_event_record(e0))
    for _ in range(n):
        fn()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float()
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return 1e3 * ms.value / n


fwd = lambda: L.check(lib.lg_head_fwd_f32(x.ptr, hidden, 1, w.ptr, b.ptr, tgt.ptr, y.ptr, err.ptr, row_loss.ptr, rows, hidden, outs))   # noqa: E731
fwd_ahead = lambda: L.check(lib.lg_head_fwd_grad_f32(x.ptr, hidden, 1, w.ptr, b.ptr, tgt.ptr, y.ptr, err.ptr, row_loss.ptr, dx.ptr, gpre.ptr, rows, hidden, outs))   # noqa: E731
bwd = lambda: L.check(lib.lg_head_bwd_f32(x.ptr, hidden, 1, g.ptr, w.ptr, dx.ptr, gpre.ptr, dw.ptr, 0, db.ptr, 0, rows, hidden, outs, row_loss.ptr, loss.ptr))    # noqa: E731
empty = lambda: L.check(lib.lg_counter_add_i64(HipTensor._new_step_counter(0).ptr if False else cnt.ptr, 1))                              # noqa: E731
cnt = HipTensor.from_numpy(np.zeros(2, np.int64), requires_grad=False)
print("LG_HEAD_FWD=%s  rows %d hidden %d outs %d:  empty kernel %.2f us   head_fwd %.2f us   head_fwd + gradients ahead %.2f us   head_bwd %.2f us"
      % (os.environ.get("LG_HEAD_FWD", "regs"), rows, hidden, outs, timed(empty), timed(fwd), timed(fwd_ahead), timed(bwd)))
