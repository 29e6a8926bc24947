"""Time lg_cross_entropy_f32 on tiny-BERT's (1024, 30522) logits (HIP events, 20 back-to-back launches) - once per
LG_CE_HELD setting to compare the kernels."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgrad_amd import HipTensor                          # noqa: E402
from lightgrad_amd.autograd.hip import lib as L              # noqa: E402

lib = L.lib()
rng = np.random.RandomState(0)
rows, cols = 1024, 30522
logits = HipTensor.from_numpy(rng.uniform(-8, 8, (rows, cols)).astype(np.float32))
labels = HipTensor.from_numpy(rng.randint(0, cols, rows).astype(np.int64), requires_grad=False)
dl, nll = HipTensor.empty((rows, cols)), HipTensor.empty((rows,))


def event():
    e = ctypes.c_void_p()
    L.check(lib.lg_event_create(ctypes.byref(e)))
    return e


def run():
    L.check(lib.lg_cross_entropy_f32(logits.ptr, labels.ptr, 8, dl.ptr, nll.ptr, rows, cols))


for _ in range(3):
    run()
best = 1e9
for _ in range(5):
    e0, e1 = event(), event()
    L.check(lib.lg_event_record(e0))
    for _ in range(20):
        run()
    L.check(lib.lg_event_record(e1))
    ms = ctypes.c_float()
    L.check(lib.lg_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    best = min(best, 1e3 * ms.value / 20)
print("cross_entropy (1024, 30522): %.2f us per launch, %.2f TB/s of logits read + gradient written   [LG_CE_HELD=%s ]"
      % (best, 2 * rows * cols * 4 / best * 1e-6, os.environ.get("LG_CE_HELD", "1")))
